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


# ---- BASELINE configs[2]: the pathogenic catalog with expanded alleles (HMM stress) -------------------------------------------
# The 56 motif sets of the reference's pathogenic catalog (repeats/pathogenic_repeats.hg38.bed: ID, MOTIFS, STRUC; kept as data in
# trgt_amd/data/pathogenic_motif_sets.json) with, per locus, one allele of 10-40 motif units and one expanded allele whose length
# is log-uniform in [500, max_allele_bp]; 30 reads per locus with the error model of SURVEY.md Appendix E.  Host-side numpy (70
# loci): the generator of the other configs (trgt_synth_generate, host C++) only knows single synthetic motif sets.
def _mutate_np(rng, seq, sub, dele, ins):
    a = np.frombuffer(seq, np.uint8).copy()
    n = len(a)
    if n == 0:
        return seq
    r = rng.random(n)
    s = r < sub
    if s.any():
        bases = np.frombuffer(b"ACGT", np.uint8)
        code = np.zeros(256, np.int64)
        code[bases] = np.arange(4)
        a[s] = bases[(code[a[s]] + rng.integers(1, 4, size=int(s.sum()))) % 4]  # another base
    keep = ~((r >= sub) & (r < sub + dele))
    a = a[keep]
    q = rng.random(len(a)) < ins
    if q.any():
        pos = np.nonzero(q)[0] + 1
        a = np.insert(a, pos, np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=len(pos))])
    return a.tobytes()


def generate_cfg3(n_loci=70, first_locus=0, seed=20250509, max_allele_bp=10000, reads_per_locus=30, flank_len=250, context_len=250,
                  truncate_rate=0.10):
    """Returns the same dict of ABI arrays as generate() (through locus.pack) for loci [first_locus, first_locus + n_loci)."""
    import json
    import os

    from . import locus as _locus
    sets = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "pathogenic_motif_sets.json")))["loci"]
    bases = np.frombuffer(b"ACGT", np.uint8)
    loci = []
    for li in range(first_locus, first_locus + n_loci):
        rng = np.random.default_rng([seed, 3, li])
        s = sets[li % len(sets)]
        motifs = [m.encode() for m in s["motifs"]]

        def fill(m):
            return bytes(b if b != ord("N") else int(bases[rng.integers(0, 4)]) for b in m)

        def allele(total):
            out = bytearray()
            per = max(1, total // len(motifs))
            for m in motifs:
                run = bytearray()
                while len(run) < per:
                    u = bytearray(fill(m))
                    if rng.random() < 0.01:
                        u[int(rng.integers(0, len(u)))] = int(bases[rng.integers(0, 4)])
                    run += u
                out += run
            return bytes(out)

        mean_len = sum(len(m) for m in motifs) / len(motifs)
        a_short = allele(int(int(rng.integers(10, 41)) * mean_len))
        a_long = allele(int(np.exp(rng.uniform(np.log(500.0), np.log(float(max_allele_bp))))))
        rnd = lambda n: bases[rng.integers(0, 4, size=n)].tobytes()
        lf, rf, lc, rc = rnd(flank_len), rnd(flank_len), rnd(context_len), rnd(context_len)
        reads = []
        for i in range(reads_per_locus):
            r = _mutate_np(rng, lc + lf + (a_short if i % 2 else a_long) + rf + rc, 5e-4, 2.5e-4, 2.5e-4)
            if rng.random() < truncate_rate:  # a read that ends (or starts) inside the locus
                cut = int(rng.integers(context_len + flank_len // 2, max(context_len + flank_len // 2 + 1, len(r) - context_len)))
                r = r[:cut] if rng.random() < 0.5 else r[len(r) - cut:]
            reads.append(r)
        loci.append(dict(left_flank=lf, right_flank=rf, tr=a_short, motifs=motifs, reads=reads, ploidy=2))
    return _locus.pack(loci)
