import sys, os
sys.path.insert(0, "/root/repo")
import torch
from trgt_amd import locus, synth, _lib
b = synth.generate(2000, first_locus=0, config=5)
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b)
tctx = _lib.context_with_env(TRGT_TIMELINE=1)
tctx.timing_enable(True)
for i in range(5):
    print("---- call", i, file=sys.stderr)
    locus.run_batch(b, locus.Params(host_threads=8), tctx, out, flank_dev=fd, reads_dev=rd)
