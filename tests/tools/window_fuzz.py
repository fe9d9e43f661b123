"""Seeded fuzz aimed at the seeded windows of the flank fallback alignments (trgt_amd/csrc/spans.hip): loci whose reads carry the flanks
with a random number of substitutions (0-12), insertions and deletions of random lengths (1-24), second copies of a flank at random
distances, periodic flanks, flanks clipped by the ends of the read.  Every locus goes through trgt_locus_batch (reads in HBM) and
through the oracle; records are compared as in parity_sweep.py.

    python tests/tools/window_fuzz.py [loci_per_round=3000] [rounds=6] [seed=1] [x,o,e]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from trgt_amd import locus, _lib
from oracle import binding as oracle
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from parity_sweep import gpu_records

LUT = np.frombuffer(b"ACGT", np.uint8)


def rnd(rng, n):
    return LUT[rng.integers(0, 4, n)]


def edit_flank(rng, f):
    """the flank with a random handful of edits"""
    a = f.copy()
    kind = rng.integers(0, 8)
    nsub = int(rng.integers(0, 13)) if kind < 5 else int(rng.integers(0, 4))
    if nsub:
        pos = rng.choice(len(a), nsub, replace=False)
        a[pos] = rnd(rng, nsub)   # (a quarter of them leave the base as it is)
    if kind in (5, 6, 7) or rng.random() < 0.15:
        p = int(rng.integers(1, len(a) - 1)); g = int(rng.integers(1, 25))
        if rng.random() < 0.5:
            a = np.concatenate([a[:p], rnd(rng, g), a[p:]])
        else:
            a = np.concatenate([a[:p], a[p + g:]])
    return a


def make_locus(rng):
    periodic = rng.random() < 0.15
    if periodic:
        unit = rnd(rng, int(rng.integers(1, 60)))
        lf = np.tile(unit, 250 // len(unit) + 1)[:250]
    else:
        lf = rnd(rng, 250)
    rf = rnd(rng, 250)
    tr = np.tile(rnd(rng, 3), 20)
    reads = []
    for i in range(12):
        l, r = edit_flank(rng, lf), edit_flank(rng, rf)
        left, right = rnd(rng, int(rng.integers(200, 320))), rnd(rng, int(rng.integers(200, 320)))
        mode = rng.integers(0, 10)
        if mode == 0:      # a second copy of the left flank in the left context
            cpy = edit_flank(rng, lf)
            left = np.concatenate([left[:40], cpy, left[40:40 + int(rng.integers(0, 60))]])
        elif mode == 1:    # flank clipped by the start of the read
            left = rnd(rng, int(rng.integers(0, 45)))
            right = rnd(rng, int(rng.integers(400, 520)))
        elif mode == 2:    # ... by its end
            right = rnd(rng, int(rng.integers(0, 45)))
            left = rnd(rng, int(rng.integers(400, 520)))
        read = np.concatenate([left, l, tr, r, right])[:1230]
        reads.append(read.tobytes())
    reads.append(np.concatenate([rnd(rng, 330), lf, tr, rf, rnd(rng, 330)]).tobytes())   # keeps the others above the short-read cut-off
    return dict(left_flank=lf.tobytes(), right_flank=rf.tobytes(), tr=tr.tobytes(), motifs=[tr[:3].tobytes()], ploidy=2, reads=reads)


def main():
    n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    scoring = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (2, 5, 1)
    threads = min(os.cpu_count() or 1, 128)
    ctx = _lib.context_with_env(TRGT_WFA_DEBUG=1)
    bad = total = 0
    for r in range(rounds):
        rng = np.random.default_rng(seed * 1000 + r)
        b = locus.pack([make_locus(rng) for _ in range(n_loci)])
        out = locus.run_batch(b, locus.Params(aln_scoring=scoring), ctx, flank_dev=torch.from_numpy(b["flank_blob"]).cuda(), reads_dev=torch.from_numpy(b["read_blob"]).cuda())
        got = gpu_records(b, out)
        ref = oracle.locus_records(b, 0, n_loci, threads, scoring=scoring)
        nb = sum(g != x for g, x in zip(got, ref))
        if nb:
            l = next(i for i in range(n_loci) if got[i] != ref[i])
            print("MISMATCH round %d locus %d\n  gpu    %s\n  oracle %s" % (r, l, got[l][:500], ref[l][:500]))
        bad += nb
        total += n_loci
        print("[window fuzz] round %d: %d loci, %d mismatches" % (r, n_loci, nb), flush=True)
    print("RESULT window fuzz: loci=%d mismatches=%d seed=%d scoring=%s" % (total, bad, seed, scoring))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
