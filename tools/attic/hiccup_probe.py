#!/usr/bin/env python3
"""developer helper: many consecutive calls on the bench workload; prints the stage split of every call slower than 1.2x the median."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trgt_amd import locus, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
b = synth.generate(10000, first_locus=0)
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b); ctx = _lib.Context(0)
params = locus.Params(host_threads=os.cpu_count())
rows = []
for i in range(n):
    t0 = time.perf_counter(); locus.run_batch(b, params, ctx, out, flank_dev=fd, reads_dev=rd); dt = (time.perf_counter() - t0) * 1e3
    rows.append((dt, [round(float(v) / 1e6, 2) for v in out.stats[4:14]]))
med = sorted(r[0] for r in rows[5:])[len(rows[5:]) // 2]
print("median %.2f ms" % med, "stats = [waitA, B, C, host, total, setup, +uploads, +enqueueA, select, gather]")
for i, (dt, st) in enumerate(rows):
    if i >= 5 and dt > 1.2 * med: print(i, "%.2f ms" % dt, st)
