#!/usr/bin/env python3
"""developer helper: per-stage wall times of trgt_locus_batch on the bench workload (stats[4..15], ms)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trgt_amd import locus, synth
b = synth.generate(10000, first_locus=0)
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b)
for i in range(4):
    locus.run_batch(b, outputs=out, flank_dev=fd, reads_dev=rd)
s = out.stats
names = ["A_wait", "B_consensus", "C_hmm", "host_glue", "total", "setup", "+uploads", "+enqueueA", "select", "gather", "front", "back"]
print({n: round(float(v) / 1e6, 2) for n, v in zip(names, s[4:16])})
