import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from trgt_amd import locus, synth, _lib
from trgt_amd.driver import ChunkDriver
cfg = int(sys.argv[1]); k = int(sys.argv[2])
n = {2: 10000, 4: 10000, 5: 2000}[cfg]
b = synth.generate(n, first_locus=0, config=cfg)
rd, fd = torch.from_numpy(b["read_blob"]).cuda(), torch.from_numpy(b["flank_blob"]).cuda()
params = locus.Params(host_threads=8)
steps = 60
for rep in range(3):
    drv = ChunkDriver(devices=[0] * k, params=params)
    outs = [locus.BatchOutputs(b) for _ in range(k)]
    wk = lambda w: dict(outputs=outs[w], flank_dev=fd, reads_dev=rd)
    drv.run([b] * (3 * k), worker_kwargs=wk); torch.cuda.synchronize()
    t0 = time.perf_counter(); drv.run([b] * steps, worker_kwargs=wk); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("python driver: %.0f loci/s" % (n * steps / dt)); drv.close()
    pool = _lib.Pool([0] * k)
    many = lambda m: locus.run_many(pool, [b] * m, params, outs, flank_dev=fd, reads_dev=rd, out_per_context=True)
    many(3 * k); torch.cuda.synchronize()
    t0 = time.perf_counter(); many(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("native pool:   %.0f loci/s" % (n * steps / dt)); pool.close()
