"""Host-side timeline (TRGT_TIMELINE=1, ms since the call started) of the fourth call of a cfg5 batch: the cluster genotyper's rounds."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trgt_amd import locus, synth, _lib
b = synth.generate(2000, first_locus=0, config=5)
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b)
ctx = _lib.context_with_env(TRGT_TIMELINE=1)
for i in range(4):
    if i == 3:
        sys.stderr.write("==== call 4\n")
    locus.run_batch(b, locus.Params(), ctx, out, flank_dev=fd, reads_dev=rd)
