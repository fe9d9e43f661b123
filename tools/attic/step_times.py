#!/usr/bin/env python3
"""developer helper: wall time of consecutive trgt_locus_batch calls on the bench workload (optionally with kernel timing on)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trgt_amd import locus, synth, _lib
timing = len(sys.argv) > 1 and sys.argv[1] == "timing"
b = synth.generate(10000, first_locus=0)
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b); ctx = _lib.Context(0)
params = locus.Params(host_threads=os.cpu_count())
ts = []
for i in range(int(os.environ.get('N_STEPS', '14'))):
    if i == 2 and timing: ctx.timing_enable(True); ctx.timing_reset()
    if i == 8 and timing and len(sys.argv) > 2: ctx.timing_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter(); locus.run_batch(b, params, ctx, out, flank_dev=fd, reads_dev=rd); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("timing" if timing else "plain", " ".join("%.1f" % t for t in ts))
