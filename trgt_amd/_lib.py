"""ctypes loader for trgt_amd/libtrgt_hip.so (the C ABI declared in include/trgt_hip.h).

The library is built in-tree by `make -C trgt_amd/csrc` (hipcc, gfx950).  Loading fails loudly when
it is missing; creating a context fails loudly when no gfx950 GPU is visible -- there is no fallback.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TRGT_HIP_LIB: another build of the same library, e.g. the developer build `make -C trgt_amd/csrc DEV=1` -> libtrgt_hip_dev.so)
_SO = os.environ.get("TRGT_HIP_LIB") or os.path.join(_HERE, "libtrgt_hip.so")
_LIB = None
_CTX = {}


class TrgtHipError(RuntimeError):
    pass


class WfaParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "metric", "mismatch", "gap_open1", "gap_ext1", "gap_open2", "gap_ext2", "span", "pattern_begin_free",
        "pattern_end_free", "text_begin_free", "text_end_free", "scope", "memory_mode", "heuristic",
        "h_min_wavefront_length", "h_max_distance_threshold", "h_steps_between_cutoffs", "bialign_min_score",
        "bialign_min_length")]


class SpanParams(C.Structure):
    _fields_ = [("flank_len", C.c_int32), ("min_flank_id_frac", C.c_double), ("mism", C.c_int32), ("gapo", C.c_int32),
                ("gape", C.c_int32)]


class LocusParams(C.Structure):
    _fields_ = [("flank_len", C.c_int32), ("min_flank_id_frac", C.c_double), ("max_depth", C.c_int32),
                ("mism", C.c_int32), ("gapo", C.c_int32), ("gape", C.c_int32), ("host_threads", C.c_int32),
                ("min_read_qual", C.c_double)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("config", C.c_int32), ("reads_per_locus", C.c_int32), ("context_len", C.c_int32),
                ("flank_len", C.c_int32), ("max_allele_bp", C.c_int32), ("sub_rate", C.c_double), ("del_rate", C.c_double),
                ("ins_rate", C.c_double), ("stutter_rate", C.c_double), ("truncate_rate", C.c_double)]


class SynthBatch(C.Structure):
    _fields_ = ([(n, C.c_int64) for n in ("n_loci", "n_reads", "n_motifs")] +
                [(n, C.c_uint64) for n in ("flank_bytes", "tr_bytes", "motif_bytes", "read_bytes")] +
                [(n, C.c_void_p) for n in ("flank_blob", "lf_off", "lf_len", "rf_off", "rf_len", "tr_blob", "tr_off", "tr_len",
                                           "motif_blob", "motif_off", "set_motif_begin", "ploidy", "locus_read_begin",
                                           "read_blob", "read_off", "read_len", "true_allele_len", "read_hap",
                                           "read_truncated", "genotyper")])


_VP = C.c_void_p


class LocusBatchIn(C.Structure):
    _fields_ = [("n_loci", C.c_int64)] + [(n, _VP) for n in (
        "flank_blob", "lf_off", "lf_len", "rf_off", "rf_len", "tr_blob", "tr_off", "tr_len", "motif_blob", "motif_off",
        "set_motif_begin", "ploidy", "locus_read_begin", "read_blob", "read_off", "read_len", "genotyper", "read_qual",
        "hp_tag", "start_offset", "end_offset", "mismatch_offsets", "mismatch_off")] + [("read_encoding", C.c_int32)]


class LocusBatchOut(C.Structure):
    _fields_ = [(n, _VP) for n in (
        "span_start", "span_end", "n_alleles", "allele_blob", "allele_off", "allele_cap", "allele_len", "ci",
        "num_spanning", "classification", "read_rank", "spans3", "span_off", "n_spans", "motif_counts", "count_off",
        "purity", "stats", "gt_size", "flipped")]


EXPORTS = [
    "trgt_inflate_raw", "trgt_inflate_blocks", "trgt_deflate_blocks",
    "trgt_hip_abi_version", "trgt_hip_create", "trgt_hip_destroy", "trgt_hip_last_error", "trgt_hip_set_stream",
    "trgt_hip_set_workspace_limit", "trgt_hip_timing_enable", "trgt_hip_timing_reset", "trgt_hip_timing_get",
    "trgt_wfa_default_params", "trgt_wfa_batch", "trgt_flank_filter_batch", "trgt_find_spans_batch", "trgt_hmm_batch", "trgt_hmm_path_capacity", "trgt_hmm_models_check",
    "trgt_locus_batch", "trgt_locus_batch_submit", "trgt_locus_batch_wait", "trgt_locus_default_params", "trgt_reads_pack_bam4",
    "trgt_hip_pool_create", "trgt_hip_pool_destroy", "trgt_hip_pool_size", "trgt_hip_pool_context", "trgt_hip_pool_last_error", "trgt_locus_batch_many",
    "trgt_ingest_open", "trgt_ingest_close", "trgt_ingest_last_error", "trgt_ingest_default_params", "trgt_ingest_batch_from_catalog", "trgt_ingest_free", "trgt_ingest_device_stats", "trgt_writer_device_stats",
    "trgt_ingest_header_text", "trgt_ingest_n_contigs", "trgt_ingest_contig_name", "trgt_ingest_contig_length",
    "trgt_writer_default_params", "trgt_writer_open", "trgt_writer_write", "trgt_writer_close", "trgt_writer_last_error",
    "trgt_cigar_ref_len", "trgt_cigar_query_len", "trgt_cigar_total_query_len", "trgt_read_mismatch_offsets", "trgt_read_meth", "trgt_read_clip_to_region",
    "trgt_read_clip_bases", "trgt_median_i32", "trgt_synth_default_params", "trgt_synth_generate", "trgt_synth_free",
]


def build_extension(force=False, verbose=False, dev=False):
    """Compile every HIP translation unit for gfx950 into trgt_amd/libtrgt_hip.so (hipcc cross-compiles on CPU).  dev=True: the
    developer library trgt_amd/libtrgt_hip_dev.so (make DEV=1: result-changing switches compiled in), returned instead."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"] + (["-B"] if force else []) + (["DEV=1"] if dev else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise TrgtHipError("building libtrgt_hip.so failed")
    return os.path.join(_HERE, "libtrgt_hip_dev.so") if dev else _SO


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise TrgtHipError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the TRGT hot path)" % _SO)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.
        # Importing torch first makes our library bind to that already-loaded runtime instead of pulling a
        # second copy from /opt/rocm (two runtimes in one process leave the later one without a GPU).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_SO)
        missing = [n for n in EXPORTS if not hasattr(L, n)]
        if missing:
            raise TrgtHipError("%s does not export %s (stale build?)" % (_SO, ", ".join(missing)))
        L.trgt_hip_last_error.restype = C.c_char_p
        L.trgt_hip_last_error.argtypes = [_VP]
        L.trgt_hip_create.argtypes = [C.c_int, C.POINTER(_VP)]
        L.trgt_hip_destroy.argtypes = [_VP]
        L.trgt_hip_destroy.restype = None
        L.trgt_hip_set_stream.argtypes = [_VP, _VP]
        L.trgt_hip_set_workspace_limit.argtypes = [_VP, C.c_uint64]
        L.trgt_hip_timing_enable.argtypes = [_VP, C.c_int]
        L.trgt_hip_timing_reset.argtypes = [_VP]
        L.trgt_hip_timing_get.argtypes = [_VP, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.trgt_hmm_path_capacity.restype = C.c_uint64
        L.trgt_hmm_path_capacity.argtypes = [C.c_uint32, C.c_uint32]
        L.trgt_hmm_batch.argtypes = [_VP, C.c_int32, _VP, _VP, _VP, C.c_int64] + [_VP] * 15
        L.trgt_hmm_models_check.argtypes = [_VP, C.c_int32, _VP, _VP, _VP, _VP]
        L.trgt_wfa_batch.argtypes = [_VP, _VP, C.c_int64] + [_VP] * 15
        L.trgt_flank_filter_batch.argtypes = [_VP, _VP, C.c_int64] + [_VP] * 5 + [C.c_int32, C.c_int32] + [_VP] * 4
        L.trgt_find_spans_batch.argtypes = [_VP, _VP, C.c_int64] + [_VP] * 13
        L.trgt_locus_batch.argtypes = [_VP, _VP, _VP, _VP]
        L.trgt_reads_pack_bam4.argtypes = [_VP, C.c_int64, _VP, _VP, _VP, _VP]
        L.trgt_reads_pack_bam4.restype = C.c_int64
        L.trgt_locus_batch_submit.argtypes = [_VP, _VP, _VP, _VP, C.POINTER(C.c_int64)]
        L.trgt_locus_batch_wait.argtypes = [_VP, C.c_int64]
        L.trgt_locus_default_params.argtypes = [_VP]
        L.trgt_locus_default_params.restype = None
        L.trgt_synth_generate.argtypes = [_VP, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.POINTER(SynthBatch))]
        L.trgt_synth_free.argtypes = [C.POINTER(SynthBatch)]
        L.trgt_synth_free.restype = None
        L.trgt_synth_default_params.argtypes = [_VP, C.c_int]
        L.trgt_synth_default_params.restype = None
        L.trgt_wfa_default_params.argtypes = [_VP]
        L.trgt_wfa_default_params.restype = None
        _LIB = L
    return _LIB


class Context:
    """One GPU + one HIP stream (mirrors the thread_local aligners of src/commands/genotype.rs:94-103)."""

    def __init__(self, device=0, creator=None):
        # creator: the library whose trgt_hip_create reads the planner knobs (the developer build reads all of them: context_with_env);
        # every other call goes through the release library, whose code honours the knobs a context carries
        h = _VP()
        L = creator or lib()
        rc = L.trgt_hip_create(int(device), C.byref(h))
        if rc != 0:
            raise TrgtHipError("trgt_hip_create(device=%d) failed (%d): %s" % (device, rc, L.trgt_hip_last_error(None).decode()))
        self.handle = h
        self.device = device

    def check(self, rc):
        if rc != 0:
            raise TrgtHipError("libtrgt_hip error %d: %s" % (rc, lib().trgt_hip_last_error(self.handle).decode()))

    def set_stream(self, stream_ptr):
        self.check(lib().trgt_hip_set_stream(self.handle, _VP(stream_ptr)))

    def timing_enable(self, on=True):
        self.check(lib().trgt_hip_timing_enable(self.handle, int(on)))

    def timing_reset(self):
        self.check(lib().trgt_hip_timing_reset(self.handle))

    def timing_get(self, kernel):
        ms, n, cells = C.c_double(), C.c_int64(), C.c_int64()
        self.check(lib().trgt_hip_timing_get(self.handle, kernel, C.byref(ms), C.byref(n), C.byref(cells)))
        return ms.value, n.value, cells.value

    def close(self):
        if self.handle:
            lib().trgt_hip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pool:
    """trgt_hip_pool: several contexts (devices[i] = ordinal of context i, ordinals may repeat) behind one queue of batches."""

    def __init__(self, devices):
        L = lib()
        L.trgt_hip_pool_create.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]
        L.trgt_hip_pool_destroy.argtypes = [C.c_void_p]
        L.trgt_hip_pool_destroy.restype = None
        L.trgt_hip_pool_context.argtypes = [C.c_void_p, C.c_int32]
        L.trgt_hip_pool_context.restype = C.c_void_p
        L.trgt_hip_pool_last_error.argtypes = [C.c_void_p]
        L.trgt_hip_pool_last_error.restype = C.c_char_p
        L.trgt_locus_batch_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        dev = (C.c_int32 * len(devices))(*devices)
        self.handle = C.c_void_p()
        rc = L.trgt_hip_pool_create(dev, len(devices), C.byref(self.handle))
        if rc != 0:
            raise TrgtHipError("trgt_hip_pool_create failed: %d %s" % (rc, L.trgt_hip_last_error(None).decode()))
        self.n = len(devices)
        # the pool owns its contexts: views that never destroy them
        self.contexts = []
        for i in range(self.n):
            c = Context.__new__(Context)
            c.handle = C.c_void_p(L.trgt_hip_pool_context(self.handle, i))
            c.device = devices[i]
            c.close = lambda: None
            self.contexts.append(c)

    def run_many(self, params_struct, cins, couts, out_per_context=False):
        """cins / couts: lists of LocusBatchIn / LocusBatchOut structures (couts: one per batch, or one per context)"""
        n = len(cins)
        pin = (C.c_void_p * n)(*[C.addressof(x) for x in cins])
        pout = (C.c_void_p * len(couts))(*[C.addressof(x) for x in couts])
        ran = (C.c_int32 * max(n, 1))()
        rc = lib().trgt_locus_batch_many(self.handle, C.byref(params_struct), n, pin, pout, 1 if out_per_context else 0, ran)
        if rc != 0:
            raise TrgtHipError("trgt_locus_batch_many error %d: %s" % (rc, lib().trgt_hip_pool_last_error(self.handle).decode()))
        return list(ran[:n])

    def close(self):
        if self.handle:
            lib().trgt_hip_pool_destroy(self.handle)
            self.handle = C.c_void_p()
            self.contexts = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# the planner switches the RELEASE library reads (tests/test_abi_exports.py pins this list against `strings libtrgt_hip.so`); every other
# TRGT_* switch -- settled A/Bs, probes -- is read only by the developer build (make -C trgt_amd/csrc DEV=1 -> libtrgt_hip_dev.so)
RELEASE_KNOBS = frozenset("TRGT_" + k for k in (
    "CLUSTER_ARENA_KB HEAVY_BAND HMM_NO_DEDUPE INFLATE_COMPILER_LOOP HMM_NO_LONG_TB HMM_NO_PPL HOST_CLUSTER HOST_GENOTYPER HOST_REPAIR INGEST_TRACE MALLOC_TUNE NO_HAMMING "
    "NO_INDEL_SHORTCUT NO_LONG_FILTER NO_ZERO_ARENA POLL_NAP_US POLL_SPIN_US POLL_WAIT REPAIR_MAX_SEG STAGE_LOCK TIMELINE WFA_DEBUG WFA_NO_EARLY "
    "WFA_NO_FILTER WFA_NO_LEAN WFA_NO_WINDOW WRITER_TRACE").split())
_DEV_SO = os.path.join(_HERE, "libtrgt_hip_dev.so")
_DEV_LIB = None


def dev_lib():
    """The developer build (reads every planner switch), used only to CREATE contexts; None when it is not built."""
    global _DEV_LIB
    if _DEV_LIB is None and os.path.exists(_DEV_SO):
        L = C.CDLL(_DEV_SO)
        L.trgt_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.trgt_hip_last_error.argtypes = [C.c_void_p]
        L.trgt_hip_last_error.restype = C.c_char_p
        _DEV_LIB = L
    return _DEV_LIB


def build_dev_extension():
    """make -C trgt_amd/csrc DEV=1 (objects under csrc/dev/): libtrgt_hip_dev.so"""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "DEV=1", "-j8"], stdout=subprocess.DEVNULL)
    return _DEV_SO


def context_with_env(device=0, **env):
    """A NEW context created while the given TRGT_* planner knobs are set in the environment: the library reads them once, in
    trgt_hip_create (none of them changes a result; the parity tests use them to pin every planner path).  A switch outside
    RELEASE_KNOBS is read by the developer build only: the context is then created by libtrgt_hip_dev.so (pytest skips when it is not
    built) and used through the release library like any other."""
    creator = None
    if any(k not in RELEASE_KNOBS for k in env):
        creator = dev_lib()
        if creator is None:
            import pytest
            pytest.skip("developer switches %s need trgt_amd/libtrgt_hip_dev.so (make -C trgt_amd/csrc DEV=1)" % sorted(k for k in env if k not in RELEASE_KNOBS))
    old = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            os.environ[k] = str(v)
        return Context(device, creator)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def context(device=0):
    """Process-wide context per device."""
    if device not in _CTX:
        _CTX[device] = Context(device)
    return _CTX[device]


class DevPtr:
    """A raw device address (trgt_ingest_batch::read_blob_dev) wherever a torch tensor in HBM is accepted (reads_dev=...); `owner` keeps
    the native object that owns the memory alive."""

    def __init__(self, addr, owner=None):
        self.addr, self.owner = int(addr), owner

    def data_ptr(self):
        return self.addr


def ptr(a):
    """void* of a numpy array (host) or a torch tensor (host or HBM); None -> NULL."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return _VP(a.data_ptr())
    return a.ctypes.data_as(_VP)
