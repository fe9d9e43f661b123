"""developer probe: (a) back-to-back uploads of the cfg2 read blob from pinned memory, (b) the same next to the resident compute loop of
another thread / context: what does each lose?"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trgt_amd import locus, synth, _lib
b = synth.generate(10000, first_locus=0, config=2)
pin = torch.from_numpy(b["read_blob"]).pin_memory()
dst = [torch.empty_like(pin, device="cuda") for _ in range(2)]
s = torch.cuda.Stream()
def uploads(n):
    with torch.cuda.stream(s):
        for i in range(n):
            dst[i % 2].copy_(pin, non_blocking=True)
    s.synchronize()
uploads(3)
t0 = time.perf_counter(); uploads(20); dt = time.perf_counter() - t0
print("uploads alone: %.2f ms per %d MB = %.1f GB/s" % (1e3 * dt / 20, pin.numel() >> 20, pin.numel() * 20 / dt / 1e9))
rd, fd = torch.from_numpy(b["read_blob"]).cuda(), torch.from_numpy(b["flank_blob"]).cuda()
ctx = _lib.Context(0); out = locus.BatchOutputs(b); params = locus.Params(host_threads=8)
for _ in range(4): locus.run_batch(b, params, ctx, out, flank_dev=fd, reads_dev=rd)
t0 = time.perf_counter()
for _ in range(30): locus.run_batch(b, params, ctx, out, flank_dev=fd, reads_dev=rd)
print("compute alone: %.2f ms per step" % (1e3 * (time.perf_counter() - t0) / 30))
res = {}
def comp():
    t0 = time.perf_counter()
    for _ in range(30): locus.run_batch(b, params, ctx, out, flank_dev=fd, reads_dev=rd)
    res["c"] = 1e3 * (time.perf_counter() - t0) / 30
def upl():
    t0 = time.perf_counter(); uploads(30); res["u"] = 1e3 * (time.perf_counter() - t0) / 30
tc, tu = threading.Thread(target=comp), threading.Thread(target=upl)
tc.start(); tu.start(); tc.join(); tu.join()
print("together: compute %.2f ms per step, uploads %.2f ms each" % (res["c"], res["u"]))
