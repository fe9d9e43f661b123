"""Synthetic locus batches -- thin wrapper over trgt_synth_generate (host C++, trgt_amd/csrc/synth.hip).

The generator is deterministic and shard-local: locus i is produced from splitmix64(seed ^ (i+1)*0x9E3779B97F4A7C15), so rank
r of N can generate exactly its own loci [r*n/N, (r+1)*n/N) with no communication.
"""
import ctypes as C

import numpy as np

from . import _lib

_FIELDS = [("flank_blob", np.uint8, "flank_bytes"), ("lf_off", np.uint64, "n_loci"), ("lf_len", np.uint32, "n_loci"),
           ("rf_off", np.uint64, "n_loci"), ("rf_len", np.uint32, "n_loci"), ("tr_blob", np.uint8, "tr_bytes"),
           ("tr_off", np.uint64, "n_loci"), ("tr_len", np.uint32, "n_loci"), ("motif_blob", np.uint8, "motif_bytes"),
           ("motif_off", np.uint32, "n_motifs+1"), ("set_motif_begin", np.uint32, "n_loci+1"), ("ploidy", np.uint8, "n_loci"),
           ("locus_read_begin", np.uint64, "n_loci+1"), ("read_blob", np.uint8, "read_bytes"), ("read_off", np.uint64, "n_reads"),
           ("read_len", np.uint32, "n_reads"), ("true_allele_len", np.uint32, "2*n_loci"), ("read_hap", np.uint8, "n_reads"),
           ("read_truncated", np.uint8, "n_reads"), ("genotyper", np.uint8, "n_loci")]


def default_params(config=2, **overrides):
    p = _lib.SynthParams()
    _lib.lib().trgt_synth_default_params(C.byref(p), config)
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


def generate(n_loci, first_locus=0, config=2, threads=0, **overrides):
    """Returns a dict of numpy arrays in the trgt_locus_batch_in layout (+ ground truth)."""
    p = default_params(config, **overrides)
    h = C.POINTER(_lib.SynthBatch)()
    rc = _lib.lib().trgt_synth_generate(C.byref(p), first_locus, n_loci, threads, C.byref(h))
    if rc != 0:
        raise _lib.TrgtHipError("trgt_synth_generate failed: %d" % rc)
    b = h.contents
    out = dict(n_loci=int(b.n_loci), n_reads=int(b.n_reads), n_motifs=int(b.n_motifs), params=p)
    env = dict(n_loci=out["n_loci"], n_reads=out["n_reads"], n_motifs=out["n_motifs"], flank_bytes=int(b.flank_bytes),
               tr_bytes=int(b.tr_bytes), motif_bytes=int(b.motif_bytes), read_bytes=int(b.read_bytes))
    for name, dt, count in _FIELDS:
        n = int(eval(count, {}, env))
        addr = getattr(b, name)
        arr = np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint8)), shape=(max(n, 0) * np.dtype(dt).itemsize,))
        out[name] = arr.view(dt).copy() if n else np.zeros(0, dt)
    _lib.lib().trgt_synth_free(h)
    return out
