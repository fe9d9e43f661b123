#!/usr/bin/env python3
"""Which alignments the pre-filter spends its levels on: the expensive fallback jobs of the bench workload (reads too short to span
their locus, flank piece not found exactly), their exact penalties with early rejection off, and how many levels x diagonals each kind costs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trgt_amd import _lib, synth, wfaligner

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
b = synth.generate(n, config=2)
F = 250
MINM = int(sys.argv[2]) if len(sys.argv) > 2 else 175  # ceil(250 * 0.7): --min-flank-id-frac default
pats, txts = [], []
lrb = b["locus_read_begin"]
for l in range(n):
    lf = bytes(b["flank_blob"][int(b["lf_off"][l]):int(b["lf_off"][l]) + F])
    rf = bytes(b["flank_blob"][int(b["rf_off"][l]):int(b["rf_off"][l]) + F])
    lens = b["read_len"][int(lrb[l]):int(lrb[l + 1])]
    heavy = int(lens.max()) - (F + F // 5)
    for r in range(int(lrb[l]), int(lrb[l + 1])):
        ln = int(b["read_len"][r])
        if ln >= heavy:
            continue
        rd = bytes(b["read_blob"][int(b["read_off"][r]):int(b["read_off"][r]) + ln])
        for piece in (lf, rf):
            if rd.find(piece) < 0:
                pats.append(piece)
                txts.append(rd)
ctx = _lib.Context(0)
full = wfaligner.flank_filter_batch(pats, txts, MINM, ctx=ctx, early_reject=False)
early = wfaligner.flank_filter_batch(pats, txts, MINM, ctx=ctx, early_reject=True)
pen = -full["score"].astype(np.int64)
tl = np.array([len(t) for t in txts])
keep = full["keep"].astype(bool)
print("jobs %d (%.2f per locus), kept %d; offsets computed: no early rejection %.3g, with %.3g" % (len(pats), len(pats) / n, int(keep.sum()), full["offsets"], early["offsets"]))
work = 3.0 * pen * (tl + 1)
print("share of (penalty x text length) by final penalty, and mean text length:")
for lo, hi in ((0, 8), (8, 16), (16, 32), (32, 52), (52, 64), (64, 96), (96, 128), (128, 192), (192, 400)):
    m = (pen >= lo) & (pen < hi)
    if m.any():
        print("  penalty %3d..%3d: %6d jobs (%5.1f %%), kept %5d, work share %5.1f %%, mean text %4.0f" % (lo, hi - 1, int(m.sum()), 100 * m.mean(), int((m & keep).sum()), 100 * work[m].sum() / work.sum(), tl[m].mean()))
rej = early["score"] == np.iinfo(np.int32).min + 1
print("given up early: %d of %d" % (int(rej.sum()), len(pats)))
