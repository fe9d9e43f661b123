import sys, os
sys.path.insert(0, "/root/repo")
import torch
from trgt_amd import locus, synth, _lib
b = synth.generate(2000, first_locus=0, config=5)
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b); ctx = _lib.Context(0)
for i in range(4):
    if i == 3: os.environ["TRGT_TIMELINE"] = "1"
    locus.run_batch(b, locus.Params(), ctx, out, flank_dev=fd, reads_dev=rd)
