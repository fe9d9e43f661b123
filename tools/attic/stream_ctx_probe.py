"""developer probe: reads starting in pinned HOST memory, K contexts per GPU each running the blocking call (its own upload + compute):
python tools/stream_ctx_probe.py [config] [contexts ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trgt_amd import locus, synth, _lib
from trgt_amd.driver import ChunkDriver
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ks = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 6]
n = {2: 10000, 4: 10000, 5: 2000}[cfg]
b = synth.generate(n, first_locus=0, config=cfg)
flank_dev = torch.from_numpy(b["flank_blob"]).cuda()
params = locus.Params(host_threads=8)
for k in ks:
    drv = ChunkDriver(devices=[0] * k, params=params)
    pins = [torch.from_numpy(b["read_blob"]).pin_memory() for _ in range(k)]
    outs = [locus.BatchOutputs(b) for _ in range(k)]
    wk = lambda w: dict(outputs=outs[w], flank_dev=flank_dev, reads_dev=pins[w])
    drv.run([b] * (3 * k), worker_kwargs=wk)
    torch.cuda.synchronize()
    steps = 60
    t0 = time.perf_counter()
    drv.run([b] * steps, worker_kwargs=wk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("config %d, %d contexts, reads in pinned host memory: %.0f loci/s (%.2f ms / step)" % (cfg, k, n * steps / dt, 1e3 * dt / steps))
    drv.close()
