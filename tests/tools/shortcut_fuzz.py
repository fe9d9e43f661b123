"""Seeded fuzz aimed at the two shortcuts of the seed search (trgt_amd/csrc/spans.hip: substitution-only flanks, one-base gaps): flanks of
low complexity (homopolymer and short-period runs, tandem duplications inside the flank, wholly periodic flanks) next to random ones;
reads that carry them with exactly one / two / three substitutions, one inserted or deleted base (inside runs, at the margins of the
flank, next to a substitution), two gaps, a two-base gap, and decoy copies of the flank elsewhere in the read.  Every locus goes through
trgt_locus_batch (reads in HBM) and through the oracle; records are compared as in parity_sweep.py.

    python tests/tools/shortcut_fuzz.py [loci_per_round=4000] [rounds=8] [seed=1]
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


def make_flank(rng):
    kind = rng.integers(0, 6)
    f = rnd(rng, 250)
    if kind == 1:    # a few runs of a short unit
        for _ in range(int(rng.integers(1, 5))):
            unit = rnd(rng, int(rng.integers(1, 4)))
            n = int(rng.integers(5, 70)); at = int(rng.integers(0, 250 - n))
            f[at:at + n] = np.tile(unit, n // len(unit) + 1)[:n]
    elif kind == 2:  # a stretch of the flank duplicated in tandem
        n = int(rng.integers(12, 70)); at = int(rng.integers(0, 250 - 2 * n))
        f[at + n:at + 2 * n] = f[at:at + n]
    elif kind == 3:  # periodic all the way
        unit = rnd(rng, int(rng.integers(1, 45)))
        f = np.tile(unit, 250 // len(unit) + 1)[:250]
    elif kind == 4:  # runs at both margins
        for at in (0, 250 - 30):
            f[at:at + 30] = np.tile(rnd(rng, int(rng.integers(1, 3))), 31)[:30]
    return f


def pos_of(rng, n):
    r = rng.random()
    if r < 0.25:
        return int(rng.integers(0, 16))
    if r < 0.5:
        return int(rng.integers(n - 40, n))
    return int(rng.integers(0, n))


def sub(rng, a, p):
    a[p] = LUT[(int(np.searchsorted(LUT, a[p])) + int(rng.integers(1, 4))) % 4]


def edit(rng, f):
    a = f.copy()
    kind = int(rng.integers(0, 12))
    if kind in (0, 1):
        sub(rng, a, pos_of(rng, len(a)))
    elif kind == 2:
        for _ in range(2):
            sub(rng, a, pos_of(rng, len(a)))
    elif kind == 3:
        for _ in range(3):
            sub(rng, a, pos_of(rng, len(a)))
    elif kind in (4, 5):   # one inserted base: a copy of its neighbour half of the time
        p = pos_of(rng, len(a))
        b = a[p:p + 1] if rng.random() < 0.5 else rnd(rng, 1)
        a = np.concatenate([a[:p], b, a[p:]])
    elif kind in (6, 7):   # one deleted base
        p = pos_of(rng, len(a))
        a = np.concatenate([a[:p], a[p + 1:]])
    elif kind == 8:        # a gap and a substitution
        p = pos_of(rng, len(a))
        a = np.concatenate([a[:p], a[p + 1:]]) if rng.random() < 0.5 else np.concatenate([a[:p], rnd(rng, 1), a[p:]])
        sub(rng, a, pos_of(rng, len(a)))
    elif kind == 9:        # two gaps
        for _ in range(2):
            p = pos_of(rng, len(a))
            a = np.concatenate([a[:p], a[p + 1:]]) if rng.random() < 0.5 else np.concatenate([a[:p], rnd(rng, 1), a[p:]])
    elif kind == 10:       # a gap of two bases
        p = pos_of(rng, len(a) - 2)
        a = np.concatenate([a[:p], a[p + 2:]]) if rng.random() < 0.5 else np.concatenate([a[:p], rnd(rng, 2), a[p:]])
    return a               # (kind 11: untouched)


def make_locus(rng):
    lf, rf = make_flank(rng), make_flank(rng)
    tr = np.tile(rnd(rng, 3), 20)
    reads = []
    for i in range(12):
        l, r = edit(rng, lf), edit(rng, rf)
        left, right = rnd(rng, int(rng.integers(200, 320))), rnd(rng, int(rng.integers(200, 320)))
        mode = int(rng.integers(0, 10))
        if mode == 0:      # a decoy copy of the left flank (edited once more) in the left context
            cpy = edit(rng, edit(rng, lf))
            left = np.concatenate([left[:30], cpy, left[30:30 + int(rng.integers(0, 50))]])
        elif mode == 1:    # a decoy of the right flank, shifted copy of part of it, behind it
            right = np.concatenate([right[:20], rf[int(rng.integers(0, 100)):], right[20:60]])
        elif mode == 2:    # flank flush with the start of the read
            left = rnd(rng, 0)
            right = rnd(rng, int(rng.integers(450, 520)))
        elif mode == 3:    # ... with its end
            right = rnd(rng, 0)
            left = rnd(rng, int(rng.integers(450, 520)))
        reads.append(np.concatenate([left, l, tr, r, right])[:1230].tobytes())
    reads.append(np.concatenate([rnd(rng, 330), lf, tr, rf, rnd(rng, 330)]).tobytes())   # keeps the others above the short-read cut-off
    return dict(left_flank=lf.tobytes(), right_flank=rf.tobytes(), tr=tr.tobytes(), motifs=[tr[:3].tobytes()], ploidy=2, reads=reads)


def main():
    n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    threads = min(os.cpu_count() or 1, 128)
    ctx = _lib.Context(0)
    bad = total = settled = fallback = 0
    for r in range(rounds):
        rng = np.random.default_rng(seed * 7919 + r)
        b = locus.pack([make_locus(rng) for _ in range(n_loci)])
        out = locus.run_batch(b, locus.Params(), ctx, flank_dev=torch.from_numpy(b["flank_blob"]).cuda(), reads_dev=torch.from_numpy(b["read_blob"]).cuda())
        got = gpu_records(b, out)
        ref = oracle.locus_records(b, 0, n_loci, threads)
        nb = sum(g != x for g, x in zip(got, ref))
        if nb:
            l = next(i for i in range(n_loci) if got[i] != ref[i])
            print("MISMATCH round %d locus %d\n  gpu    %s\n  oracle %s" % (r, l, got[l][:600], ref[l][:600]))
        bad += nb
        total += n_loci
        settled += int(out.stats[21]); fallback += int(out.stats[0])
        print("[shortcut fuzz] round %d: %d loci, %d mismatches, %d of %d fallback alignments settled by the shortcuts" % (r, n_loci, nb, int(out.stats[21]), int(out.stats[0])), flush=True)
    print("RESULT shortcut fuzz: loci=%d mismatches=%d settled=%d of %d fallback alignments seed=%d" % (total, bad, settled, fallback, seed))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
