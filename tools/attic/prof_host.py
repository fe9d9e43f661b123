import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from trgt_amd import _lib, locus, synth
b = synth.generate(10000)
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
ctx = _lib.Context(0); out = locus.BatchOutputs(b); p = locus.Params(host_threads=32)
for i in range(8):
    t = time.perf_counter(); locus.run_batch(b, p, ctx, out, flank_dev=fd, reads_dev=rd); dt = time.perf_counter() - t
    s = out.stats
    print("step %.1f ms | inside %.1f: flank %.1f cons %.1f hmm %.1f host %.1f | hmm: model %.1f jobs %.1f bufs %.1f | glue: select %.1f +gather %.1f +front %.1f ; back %.1f" % (
        dt*1e3, s[8]/1e6, s[4]/1e6, s[5]/1e6, s[6]/1e6, s[7]/1e6, s[9]/1e6, s[10]/1e6, s[11]/1e6, s[12]/1e6, s[13]/1e6, s[14], s[15]))
