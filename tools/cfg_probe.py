#!/usr/bin/env python3
"""developer helper: time per trgt_locus_batch call on a synthetic batch of a given config (2, 4, 5): cfg_probe.py <config> <n_loci>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trgt_amd import locus, synth, _lib
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
b = synth.generate(n, first_locus=0, config=cfg)
print("config", cfg, "loci", n, "reads", b["n_reads"], "motifs/locus %.2f" % (b["n_motifs"] / n), "max read", int(b["read_len"].max()))
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b)
ctx = _lib.context(); ctx.timing_enable(True)
for i in range(5):
    ctx.timing_reset()
    t0 = time.perf_counter(); locus.run_batch(b, outputs=out, flank_dev=fd, reads_dev=rd); dt = time.perf_counter() - t0
    print("call %d: %.1f ms (%.0f loci/s)" % (i, dt * 1e3, n / dt), {k: round(float(v) / 1e6, 1) for k, v in zip(["A", "B", "C", "host", "total"], out.stats[4:9])},
          {k: round(ctx.timing_get(i_)[0], 2) for k, i_ in (("scan", 0), ("wfa", 1), ("hmm", 2), ("wfa_flank", 3), ("wfa_rest", 4))})
