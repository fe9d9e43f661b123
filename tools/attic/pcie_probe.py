import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from trgt_amd import locus, synth, _lib
b = synth.generate(10000, first_locus=0)
out = locus.BatchOutputs(b); ctx = _lib.Context(0)
pin = torch.from_numpy(b["read_blob"]).pin_memory()
for mode in ("pageable host reads", "pinned host reads + explicit async upload"):
    ts = []
    for i in range(6):
        t0 = time.perf_counter()
        if mode.startswith("pinned"):
            rd = pin.cuda(non_blocking=True); torch.cuda.synchronize()
            locus.run_batch(b, locus.Params(), ctx, out, reads_dev=rd)
        else:
            locus.run_batch(b, locus.Params(), ctx, out)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, " ".join("%.1f" % t for t in ts))
