"""developer probe: where does the 4-bit streaming figure lose against the resident one?  Four pool measurements on the cfg2 batch:
R0 resident ASCII; R1 the same next to a background thread that uploads 180 MB every ~4.7 ms (pure link load); R2 4-bit reads RESIDENT in
HBM (expansion and the 4-bit bookkeeping inside the call, no upload); R3 4-bit reads from pinned host memory (the streaming figure)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trgt_amd import locus, synth, _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
b = synth.generate(10000, first_locus=0, config=2)
params = locus.Params(host_threads=8)
rd, fd = torch.from_numpy(b["read_blob"]).cuda(), torch.from_numpy(b["flank_blob"]).cuda()
pk = locus.pack_bam4(b, pinned=True)
pk_dev = torch.from_numpy(pk["read_blob"]).cuda()
pool = _lib.Pool([0] * K)
outs = [locus.BatchOutputs(b) for _ in range(K)]
def rate(batch, n, **kw):
    locus.run_many(pool, [batch] * (3 * K), params, outs, flank_dev=fd, out_per_context=True, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    locus.run_many(pool, [batch] * n, params, outs, flank_dev=fd, out_per_context=True, **kw)
    torch.cuda.synchronize(); return 10000 * n / (time.perf_counter() - t0)
r0 = rate(b, N, reads_dev=rd)
print("R0 resident ASCII            %.3f M loci/s" % (r0 / 1e6))
stop = False
pin = pk["_read_blob_pinned"]; dst = [torch.empty_like(pin, device="cuda") for _ in range(2)]
s = torch.cuda.Stream(); cnt = [0]
def bg(period):
    i = 0
    with torch.cuda.stream(s):
        while not stop:
            t = time.perf_counter()
            dst[i % 2].copy_(pin, non_blocking=True); s.synchronize(); i += 1; cnt[0] += 1
            d = period - (time.perf_counter() - t)
            if d > 0: time.sleep(d)
for period in (0.0047, 0.0):
    stop = False; cnt[0] = 0
    th = threading.Thread(target=bg, args=(period,)); th.start()
    t0 = time.perf_counter(); r1 = rate(b, N, reads_dev=rd); dt = time.perf_counter() - t0
    stop = True; th.join()
    print("R1 resident + link load (period %.1f ms: %.1f GB/s)   %.3f M loci/s" % (period * 1e3, cnt[0] * pin.numel() / dt / 1e9, r1 / 1e6))
r2 = rate(pk, N, reads_dev=pk_dev)
print("R2 4-bit resident in HBM     %.3f M loci/s" % (r2 / 1e6))
r3 = rate(pk, N)
print("R3 4-bit from pinned memory  %.3f M loci/s" % (r3 / 1e6))
r0b = rate(b, N, reads_dev=rd)
print("R0 again                     %.3f M loci/s" % (r0b / 1e6))
