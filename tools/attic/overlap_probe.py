#!/usr/bin/env python3
"""developer helper: does a pinned H2D copy on a side stream overlap with the library's kernels on this box?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trgt_amd import _lib, locus, synth
b = synth.generate(10000, first_locus=0)
ctx = _lib.Context(0)
fd = torch.from_numpy(b["flank_blob"]).cuda(); rd = torch.from_numpy(b["read_blob"]).cuda()
pin = torch.from_numpy(b["read_blob"]).pin_memory()
dst = torch.empty_like(rd)
out = locus.BatchOutputs(b)
P = locus.Params(host_threads=8)
side = torch.cuda.Stream()
def compute(): locus.run_batch(b, P, ctx, out, flank_dev=fd, reads_dev=rd)
def copy():
    with torch.cuda.stream(side):
        dst.copy_(pin, non_blocking=True)
for _ in range(3): compute(); copy(); torch.cuda.synchronize()
def t(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("compute alone %.2f ms" % t(compute))
print("copy alone    %.2f ms (%.1f GB/s)" % (t(lambda: (copy(), side.synchronize())), 0), end="")
c = t(lambda: (copy(), side.synchronize())); print("  -> %.2f ms = %.1f GB/s" % (c, pin.numel() / c / 1e6))
print("copy || compute %.2f ms" % t(lambda: (copy(), compute(), side.synchronize())))
for nchunk in (8, 32, 128):
    sz = (pin.numel() + nchunk - 1) // nchunk
    def copy_chunked():
        with torch.cuda.stream(side):
            for i in range(nchunk):
                dst[i * sz:(i + 1) * sz].copy_(pin[i * sz:(i + 1) * sz], non_blocking=True)
    print("%d chunks: copy alone %.2f ms, copy || compute %.2f ms" % (nchunk, t(lambda: (copy_chunked(), side.synchronize())), t(lambda: (copy_chunked(), compute(), side.synchronize()))))
