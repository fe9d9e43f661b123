"""Native read ingestion (trgt_amd/csrc/ingest.hip through the C ABI): repeat catalog + indexed FASTA + indexed BAM -> the arrays of
trgt_locus_batch_in, i.e. analyze_tr up to clip_reads (src/trgt/workflows/tr.rs:24-35, 186-196, 268-361; locus.rs:31-98, 168-190;
reads/read.rs:55-141; reads/snp.rs:51-79; reads/clip_region.rs:19-184).  tests/pyreads.py is the Python mirror of the same steps."""
import ctypes as C

import numpy as np

from . import _lib


class IngestParams(C.Structure):
    _fields_ = [("flank_len", C.c_int32), ("max_depth", C.c_int32), ("min_read_qual", C.c_double), ("threads", C.c_int32),
                ("genotyper", C.c_int32), ("default_ploidy", C.c_int32), ("keep_bam4", C.c_int32), ("ingest_device", C.c_int32), ("inflate_waves_per_cu", C.c_int32)]


_P8, _P16, _P32, _P64, _PD, _PC = (C.POINTER(t) for t in (C.c_uint8, C.c_int16, C.c_uint32, C.c_uint64, C.c_double, C.c_char))
_PI32, _PI64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class IngestBatch(C.Structure):
    _fields_ = [("n_loci", C.c_int64), ("n_reads", C.c_int64), ("n_motifs", C.c_int64),
                ("flank_bytes", C.c_uint64), ("tr_bytes", C.c_uint64), ("motif_bytes", C.c_uint64), ("read_bytes", C.c_uint64),
                ("flank_blob", _P8), ("lf_off", _P64), ("lf_len", _P32), ("rf_off", _P64), ("rf_len", _P32),
                ("tr_blob", _P8), ("tr_off", _P64), ("tr_len", _P32),
                ("motif_blob", _P8), ("motif_off", _P32), ("set_motif_begin", _P32),
                ("ploidy", _P8), ("genotyper", _P8), ("locus_read_begin", _P64),
                ("read_blob", _P8), ("read_off", _P64), ("read_len", _P32), ("read_qual", _PD),
                ("contig_blob", _PC), ("contig_off", _P64), ("id_blob", _PC), ("id_off", _P64),
                ("struc_blob", _PC), ("struc_off", _P64), ("region_start", _PI64), ("region_end", _PI64),
                ("n_quality_filtered", _PI32), ("n_reads_seen", _PI64),
                ("qual_blob", _P8), ("name_blob", _PC), ("name_off", _P64),
                ("is_reverse", _P8), ("mapq", _P8), ("hp_tag", _P16),
                ("start_offset", _PI32), ("end_offset", _PI32),
                ("mismatch_offsets", _PI32), ("mismatch_off", _P64),
                ("meth", _P8), ("meth_off", _P64), ("has_meth", _P8),
                ("cigar", _P32), ("cigar_off", _P64), ("cigar_ref_pos", _PI64),
                ("read_bam4", _P8), ("read_bam4_off", _P64), ("read_bam4_bytes", C.c_uint64),
                ("n_skipped", C.c_int64), ("skipped_blob", _PC), ("skipped_off", _P64),
                ("read_blob_dev", C.c_void_p), ("read_blob_device", C.c_int32),
                ("owner", C.c_void_p)]


def _array(ptr, n, dtype, copy=True):
    if n <= 0:
        return np.zeros(0, dtype)
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(n) * np.dtype(dtype).itemsize,)).view(dtype)
    return a.copy() if copy else a  # (copy=False: a view of the native batch's memory)


def _strings(blob, off, n):
    o = _array(off, n + 1, np.uint64)
    raw = C.string_at(blob, int(o[-1])) if n and int(o[-1]) else b""
    return [raw[int(o[i]):int(o[i + 1])].decode() for i in range(n)]


class NativeBatch:
    """Owner of a trgt_ingest_batch (freed with the object)."""

    def __init__(self, L, handle):
        self._L, self.handle = L, handle

    def __del__(self):
        try:
            if self.handle:
                self._L.trgt_ingest_free(self.handle)
                self.handle = None
        except Exception:
            pass


class Reader:
    """An indexed BAM (<bam>.bai) and an indexed FASTA (<fasta>.fai)."""

    def __init__(self, bam_path, fasta_path):
        L = _lib.lib()
        L.trgt_ingest_open.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.trgt_ingest_close.argtypes = [C.c_void_p]
        L.trgt_ingest_close.restype = None
        L.trgt_ingest_last_error.argtypes = [C.c_void_p]
        L.trgt_ingest_last_error.restype = C.c_char_p
        L.trgt_ingest_default_params.argtypes = [C.POINTER(IngestParams)]
        L.trgt_ingest_default_params.restype = None
        L.trgt_ingest_batch_from_catalog.argtypes = [C.c_void_p, C.POINTER(IngestParams), C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.POINTER(IngestBatch))]
        L.trgt_ingest_free.argtypes = [C.POINTER(IngestBatch)]
        L.trgt_ingest_free.restype = None
        self._L = L
        self.handle = C.c_void_p()
        rc = L.trgt_ingest_open(str(bam_path).encode(), str(fasta_path).encode(), C.byref(self.handle))
        if rc != 0:
            msg = L.trgt_ingest_last_error(self.handle).decode() if self.handle else "trgt_ingest_open failed"
            if self.handle:
                L.trgt_ingest_close(self.handle)
                self.handle = C.c_void_p()
            raise _lib.TrgtHipError("trgt_ingest_open: %s" % msg)

    def device_stats(self):
        """trgt_ingest_device_stats: calls that asked for ingest_device, calls redone by the host path, reason of the last one, BGZF blocks
        through the device path, blocks its inflate kernel declined."""
        v = (C.c_int64 * 5)()
        self._L.trgt_ingest_device_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        self._L.trgt_ingest_device_stats.restype = None
        self._L.trgt_ingest_device_stats(self.handle, v)
        return dict(calls=int(v[0]), fallbacks=int(v[1]), last_reason=int(v[2]), blocks=int(v[3]), blocks_by_zlib=int(v[4]))

    def close(self):
        if self.handle:
            self._L.trgt_ingest_close(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def batch(self, bed_path, first_locus=0, max_loci=-1, keep_native=False, copy=True, read_names=True, **params):
        """Loci [first_locus, first_locus + max_loci) of the catalog as the dict of ABI arrays trgt_amd.locus.run_batch takes (the keys
        of synth.generate) plus the catalog fields and the per-read HiFiRead fields the writers need.  copy=False (with keep_native):
        the arrays are views of the native batch's memory, valid as long as batch["_native"] lives; read_names=False leaves the list of
        read names out (one Python string per read)."""
        if not copy and not keep_native:
            raise ValueError("copy=False needs keep_native=True (the views point into the native batch)")
        def _arr(ptr, n, dtype):  # (copies, or views of the native batch's memory)
            return _array(ptr, n, dtype, copy)
        p = IngestParams()
        self._L.trgt_ingest_default_params(C.byref(p))
        for k, v in params.items():
            setattr(p, k, v)
        h = C.POINTER(IngestBatch)()
        rc = self._L.trgt_ingest_batch_from_catalog(self.handle, C.byref(p), str(bed_path).encode(), first_locus, max_loci, C.byref(h))
        if rc != 0:
            raise _lib.TrgtHipError("trgt_ingest: %s" % self._L.trgt_ingest_last_error(self.handle).decode())
        b = h.contents
        nl, nr, nm = int(b.n_loci), int(b.n_reads), int(b.n_motifs)
        u8, u32, u64 = np.uint8, np.uint32, np.uint64
        out = dict(n_loci=nl, n_reads=nr, n_motifs=nm,
                   flank_blob=_arr(b.flank_blob, b.flank_bytes, u8), lf_off=_arr(b.lf_off, nl, u64), lf_len=_arr(b.lf_len, nl, u32),
                   rf_off=_arr(b.rf_off, nl, u64), rf_len=_arr(b.rf_len, nl, u32),
                   tr_blob=_arr(b.tr_blob, b.tr_bytes, u8), tr_off=_arr(b.tr_off, nl, u64), tr_len=_arr(b.tr_len, nl, u32),
                   motif_blob=_arr(b.motif_blob, b.motif_bytes, u8), motif_off=_arr(b.motif_off, nm + 1, u32),
                   set_motif_begin=_arr(b.set_motif_begin, nl + 1, u32), ploidy=_arr(b.ploidy, nl, u8), genotyper=_arr(b.genotyper, nl, u8),
                   locus_read_begin=_arr(b.locus_read_begin, nl + 1, u64),
                   read_blob=_arr(b.read_blob, b.read_bytes, u8), read_off=_arr(b.read_off, nr, u64), read_len=_arr(b.read_len, nr, u32),
                   read_qual=_arr(b.read_qual, nr, np.float64),
                   contig=_strings(b.contig_blob, b.contig_off, nl), id=_strings(b.id_blob, b.id_off, nl), struc=_strings(b.struc_blob, b.struc_off, nl),
                   region_start=_arr(b.region_start, nl, np.int64), region_end=_arr(b.region_end, nl, np.int64),
                   n_quality_filtered=_arr(b.n_quality_filtered, nl, np.int32), n_reads_seen=_arr(b.n_reads_seen, nl, np.int64),
                   qual_blob=_arr(b.qual_blob, int(_arr(b.read_len, nr, u32).sum()) if nr else 0, u8), read_name=_strings(b.name_blob, b.name_off, nr) if read_names else None,
                   is_reverse=_arr(b.is_reverse, nr, u8), mapq=_arr(b.mapq, nr, u8), hp_tag=_arr(b.hp_tag, nr, np.int16),
                   start_offset=_arr(b.start_offset, nr, np.int32), end_offset=_arr(b.end_offset, nr, np.int32),
                   mismatch_off=_arr(b.mismatch_off, nr + 1, u64), meth_off=_arr(b.meth_off, nr + 1, u64), has_meth=_arr(b.has_meth, nr, u8),
                   cigar_off=_arr(b.cigar_off, nr + 1, u64), cigar_ref_pos=_arr(b.cigar_ref_pos, nr, np.int64))
        out["mismatch_offsets"] = _arr(b.mismatch_offsets, int(out["mismatch_off"][-1]) if nr else 0, np.int32)
        out["meth"] = _arr(b.meth, int(out["meth_off"][-1]) if nr else 0, u8)
        out["cigar"] = _arr(b.cigar, int(out["cigar_off"][-1]) if nr else 0, u32)
        if len(out["read_blob"]) == 0:
            out["read_blob"] = np.zeros(1, u8)
        if b.read_bam4:  # keep_bam4=1: the reads once more as 4-bit codes; bam4_view(batch) is the batch that hands those to the GPU
            out["read_bam4"] = _arr(b.read_bam4, max(int(b.read_bam4_bytes), 1), u8)
            out["read_bam4_off"] = _arr(b.read_bam4_off, max(nr, 1), u64)
        if b.read_blob_dev:  # ingest_device: the ASCII reads are in HBM as well (owned by the native batch: keep_native=True to use them)
            out["read_blob_dev"], out["read_blob_device"] = int(b.read_blob_dev), int(b.read_blob_device)
        out["skipped"] = _strings(b.skipped_blob, b.skipped_off, int(b.n_skipped))  # "Error at BED line N: ..." per catalog line without a locus
        if keep_native:  # the writers (trgt_amd/writers.py) take the native batch itself
            out["_native"] = NativeBatch(self._L, h)
        else:
            self._L.trgt_ingest_free(h)
        return out


def inflate_blocks(ctx, streams, sizes):
    """trgt_inflate_blocks: raw DEFLATE streams (BGZF payloads) -> (list of bytes or None for a declined stream, status array).  The
    device-side stand-in for htslib's bgzf_read_block; sizes[i] is the announced inflated size of streams[i] (at most 65536)."""
    L = _lib.lib()
    n = len(streams)
    src_off = np.zeros(n, np.uint64); src_len = np.zeros(n, np.uint32); dst_off = np.zeros(n, np.uint64); dst_len = np.asarray(sizes, np.uint32)
    so = do = 0
    for i, s_ in enumerate(streams):
        src_off[i], src_len[i], dst_off[i] = so, len(s_), do
        so += (len(s_) + 15) & ~15
        do += (int(dst_len[i]) + 63) & ~63
    src = np.zeros(so + 16, np.uint8)
    for i, s_ in enumerate(streams):
        src[int(src_off[i]):int(src_off[i]) + len(s_)] = np.frombuffer(s_, np.uint8)
    dst = np.full(do + 64, 0xA5, np.uint8)
    status = np.zeros(n, np.uint8)
    L.trgt_inflate_blocks.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 7
    L.trgt_inflate_blocks.restype = C.c_int
    ctx.check(L.trgt_inflate_blocks(ctx.handle, n, src.ctypes.data, src_off.ctypes.data, src_len.ctypes.data, dst.ctypes.data, dst_off.ctypes.data, dst_len.ctypes.data, status.ctypes.data))
    out = []
    for i in range(n):
        a, k = int(dst_off[i]), int(dst_len[i])
        pad = dst[a + k:a + ((k + 63) & ~63)]
        assert (pad == 0xA5).all(), "wrote beyond a block's output"
        out.append(dst[a:a + k].tobytes() if status[i] == 1 else None)
    return out, status


def device_reads(batch):
    """The ASCII reads a batch of Reader.batch(..., ingest_device=d, keep_native=True) left in HBM, as the reads_dev argument of
    trgt_amd.locus.run_batch / run_many (nothing is uploaded then); None for a batch of the host path."""
    if not batch.get("read_blob_dev"):
        return None
    return _lib.DevPtr(batch["read_blob_dev"], batch.get("_native"))


def bam4_view(batch):
    """The batch of Reader.batch(..., keep_bam4=1) with its 4-bit reads in the place of the ASCII ones (read_encoding = TRGT_READS_BAM4):
    what trgt_amd.locus.run_batch / submit_batch / run_many upload then is half the size."""
    b = {k: v for k, v in batch.items() if k not in ("_cin", "read_bam4", "read_bam4_off")}
    b["read_blob"], b["read_off"], b["read_encoding"] = batch["read_bam4"], batch["read_bam4_off"], 1
    return b
