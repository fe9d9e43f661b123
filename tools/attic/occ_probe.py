#!/usr/bin/env python3
"""developer helper: flank-kernel time on a short-read batch (small LDS footprint) at the current TRGT_WFA_GRID_PER_CU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trgt_amd import locus, synth, _lib
ctx_len = int(sys.argv[1]) if len(sys.argv) > 1 else 250
b = synth.generate(10000, first_locus=0, context_len=ctx_len)
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b)
ctx = _lib.context(); 
for i in range(2): locus.run_batch(b, outputs=out, flank_dev=fd, reads_dev=rd)
ctx.timing_enable(True); ctx.timing_reset()
for i in range(5): locus.run_batch(b, outputs=out, flank_dev=fd, reads_dev=rd)
ms, n, cells = ctx.timing_get(3)
print("ctx", ctx_len, "max read", int(b["read_len"].max()), "grid/CU", os.environ.get("TRGT_WFA_GRID_PER_CU", "auto"), "wfa_flank %.3f ms/launch" % (ms / n), "cells/launch %.3g" % (cells / n))
