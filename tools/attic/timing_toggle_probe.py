"""Does a context stay slow after its timing events were switched off again?  (config 3, one context)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trgt_amd import locus, synth, _lib
b = synth.generate_cfg3(70)
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b)
ctx = _lib.Context(0)
def loop(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        locus.run_batch(b, locus.Params(host_threads=8), ctx, out, flank_dev=fd, reads_dev=rd)
    return 1e3 * (time.perf_counter() - t0) / n
loop(3)
print("fresh context, no events      %.1f ms / call" % loop(10))
ctx.timing_enable(True); loop(2)
print("timing events on              %.1f ms / call" % loop(10))
ctx.timing_enable(False); loop(2)
print("timing events off again       %.1f ms / call" % loop(10))
ctx.timing_reset()
print("after timing_reset            %.1f ms / call" % loop(10))
