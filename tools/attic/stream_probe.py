#!/usr/bin/env python3
"""developer helper: per-call wall time and stage split with the reads (a) resident in HBM, (b) in pinned host memory through the
blocking call, (c) in pinned host memory through submit / wait (upload of the next batch overlapped)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trgt_amd import _lib, locus, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
b = synth.generate(n, first_locus=0)
ctx = _lib.Context(0)
fd = torch.from_numpy(b["flank_blob"]).cuda(); rd = torch.from_numpy(b["read_blob"]).cuda()
pins = [torch.from_numpy(b["read_blob"]).pin_memory() for _ in range(2)]
outs = [locus.BatchOutputs(b) for _ in range(2)]
P = locus.Params(host_threads=8)
def stages(o): return {k: round(float(v) / 1e6, 2) for k, v in zip(["A", "B", "C", "host", "total"], o.stats[4:9])}
for name, fn in (("resident", lambda k: locus.run_batch(b, P, ctx, outs[0], flank_dev=fd, reads_dev=rd)),
                 ("pinned, blocking", lambda k: locus.run_batch(b, P, ctx, outs[0], flank_dev=fd, reads_dev=pins[0]))):
    for _ in range(3): fn(0)
    t0 = time.perf_counter()
    for k in range(20): fn(k)
    print("%-18s %.2f ms/call" % (name, (time.perf_counter() - t0) / 20 * 1e3), stages(outs[0]))
t = locus.submit_batch(b, P, ctx, outs[0], flank=fd, reads=pins[0])
N = 23
ts = []
for k in range(N):
    t0 = time.perf_counter()
    nxt = locus.submit_batch(b, P, ctx, outs[(k + 1) % 2], flank=fd, reads=pins[(k + 1) % 2]) if k + 1 < N else None
    t1 = time.perf_counter()
    t.wait()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1))
    t = nxt
ts = ts[3:]
print("pinned, pipelined  submit %.2f ms + wait %.2f ms per batch" % (1e3 * np.mean([a for a, _ in ts]), 1e3 * np.mean([w for _, w in ts])), stages(outs[0]))
