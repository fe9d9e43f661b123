#!/usr/bin/env python3
"""developer helper: cfg5 batch (compound / N motifs, cluster genotyper) through trgt_locus_batch: time per call, stage split,
kernel times, and the oracle's single-thread rate on the first loci."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trgt_amd import locus, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
over = dict(max_allele_bp=int(sys.argv[2])) if len(sys.argv) > 2 else {}
b = synth.generate(n, first_locus=0, config=5, **over)
print("loci", n, "reads", b["n_reads"], "motifs/locus %.1f" % (b["n_motifs"] / n), "mean allele", float(b["true_allele_len"].mean()))
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b)
ctx = _lib.context(); ctx.timing_enable(True)
for i in range(4):
    ctx.timing_reset()
    t0 = time.perf_counter(); locus.run_batch(b, outputs=out, flank_dev=fd, reads_dev=rd); dt = time.perf_counter() - t0
    print("call %d: %.1f ms (%.0f loci/s)" % (i, dt * 1e3, n / dt), {k: round(float(v) / 1e6, 1) for k, v in zip(["A", "B", "C", "host", "total"], out.stats[4:9])},
          "jobs: flank %d cons %d ed %d hmm %d" % (out.stats[0], out.stats[1], out.stats[15], out.stats[3]),
          {k: round(ctx.timing_get(i_)[0], 2) for k, i_ in (("scan", 0), ("wfa", 1), ("hmm", 2), ("wfa_flank", 3))})
if len(sys.argv) > 3:
    from oracle import binding as orc
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    t0 = time.perf_counter()
    m = int(sys.argv[3])
    lrb = b["locus_read_begin"]
    for l in range(m):
        a0, a1 = int(lrb[l]), int(lrb[l + 1])
        reads = [bytes(b["read_blob"][int(b["read_off"][r]):int(b["read_off"][r]) + int(b["read_len"][r])]) for r in range(a0, a1)]
        lf = bytes(b["flank_blob"][int(b["lf_off"][l]):int(b["lf_off"][l]) + int(b["lf_len"][l])])
        rf = bytes(b["flank_blob"][int(b["rf_off"][l]):int(b["rf_off"][l]) + int(b["rf_len"][l])])
        tr = bytes(b["tr_blob"][int(b["tr_off"][l]):int(b["tr_off"][l]) + int(b["tr_len"][l])])
        m0, m1 = int(b["set_motif_begin"][l]), int(b["set_motif_begin"][l + 1])
        motifs = [bytes(b["motif_blob"][int(b["motif_off"][k]):int(b["motif_off"][k + 1])]) for k in range(m0, m1)]
        orc.locus_analyze(lf, rf, tr, motifs, reads, genotyper=1)
    dt = time.perf_counter() - t0
    print("oracle: %d loci in %.2f s = %.0f loci/s (1 thread)" % (m, dt, m / dt))
