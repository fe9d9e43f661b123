#!/usr/bin/env python3
"""developer helper: one call with TRGT_TIMELINE=1 (host-side timeline, ms since the call started): timeline.py [config] [n_loci]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trgt_amd import locus, synth, _lib
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else {2: 10000, 3: 70, 4: 10000, 5: 2000}[cfg]
b = synth.generate_cfg3(n) if cfg == 3 else synth.generate(n, first_locus=0, config=cfg)
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b)
ctx = _lib.Context(0)
for i in range(4):
    locus.run_batch(b, locus.Params(host_threads=8), ctx, out, flank_dev=fd, reads_dev=rd)
tctx = _lib.context_with_env(TRGT_TIMELINE=1)
for i in range(3):
    print("---- call", i, file=sys.stderr)
    locus.run_batch(b, locus.Params(host_threads=8), tctx, out, flank_dev=fd, reads_dev=rd)
