"""developer probe: reads in pinned HOST memory, K contexts per GPU, each running the submit / wait pipeline (its next batch's upload next to
the compute of its current one): python tools/stream_ctx_probe2.py [config] [contexts ...]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trgt_amd import locus, synth, _lib
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ks = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
n = {2: 10000, 4: 10000, 5: 2000}[cfg]
b = synth.generate(n, first_locus=0, config=cfg)
flank_dev = torch.from_numpy(b["flank_blob"]).cuda()
params = locus.Params(host_threads=8)
for k in ks:
    ctxs = [_lib.Context(0) for _ in range(k)]
    pins = [[torch.from_numpy(b["read_blob"]).pin_memory() for _ in range(2)] for _ in range(k)]
    outs = [[locus.BatchOutputs(b) for _ in range(2)] for _ in range(k)]

    def loop(w, steps):
        t = locus.submit_batch(b, params, ctxs[w], outs[w][0], flank=flank_dev, reads=pins[w][0])
        for i in range(steps):
            nxt = locus.submit_batch(b, params, ctxs[w], outs[w][(i + 1) % 2], flank=flank_dev, reads=pins[w][(i + 1) % 2]) if i + 1 < steps else None
            t.wait()
            t = nxt

    def run(steps_each):
        th = [threading.Thread(target=loop, args=(w, steps_each)) for w in range(k)]
        for t in th: t.start()
        for t in th: t.join()

    run(3)
    torch.cuda.synchronize()
    each = max(4, 60 // k)
    t0 = time.perf_counter()
    run(each)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("config %d, %d contexts x submit/wait, reads in pinned host memory: %.0f loci/s (%.2f ms / step)" % (cfg, k, n * each * k / dt, 1e3 * dt / (each * k)))
    for c in ctxs: c.close()
