"""developer helper: host-side timeline (TRGT_TIMELINE=1) of steady-state calls of the submit / wait pipeline on 4-bit reads in pinned
host memory (one context): timeline_stream.py [n_calls]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trgt_amd import locus, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = synth.generate(10000, first_locus=0, config=2)
fd = torch.from_numpy(b["flank_blob"]).cuda()
pk = locus.pack_bam4(b, pinned=True)
outs = [locus.BatchOutputs(b) for _ in range(2)]
params = locus.Params(host_threads=8)
ctx = _lib.context_with_env(TRGT_TIMELINE=1)
t = locus.submit_batch(pk, params, ctx, outs[0], flank=fd)
marks = []
for k in range(n):
    t0 = time.perf_counter()
    nxt = locus.submit_batch(pk, params, ctx, outs[(k + 1) % 2], flank=fd) if k + 1 < n else None
    t1 = time.perf_counter()
    print("---- call %d (submit of the next took %.2f ms)" % (k, 1e3 * (t1 - t0)), file=sys.stderr, flush=True)
    t.wait()
    marks.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)))
    t = nxt
print("submit ms / wait ms per call:", " ".join("%.2f/%.2f" % m for m in marks))
