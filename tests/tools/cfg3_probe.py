#!/usr/bin/env python3
"""developer helper: cfg3-like batch (long pathogenic-like alleles) through trgt_locus_batch: time per call and a parity spot check."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trgt_amd import locus, synth
from oracle import binding as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
b = synth.generate(n, first_locus=0, max_allele_bp=int(sys.argv[2]) if len(sys.argv) > 2 else 3000)
print("reads", b["n_reads"], "max read", int(b["read_len"].max()), "mean", float(b["read_len"].mean()))
rd = torch.from_numpy(b["read_blob"]).cuda(); fd = torch.from_numpy(b["flank_blob"]).cuda()
out = locus.BatchOutputs(b)
for i in range(3):
    t0 = time.perf_counter(); locus.run_batch(b, outputs=out, flank_dev=fd, reads_dev=rd); dt = time.perf_counter() - t0
    print("call %d: %.1f ms (%.0f loci/s)" % (i, dt * 1e3, n / dt), {k: round(float(v) / 1e6, 1) for k, v in zip(["A", "B", "C", "host", "total"], out.stats[4:9])})
# parity spot check on the three loci with the fewest read bytes (the oracle is slow on long reads)
lrb = b["locus_read_begin"]
sizes = [int(b["read_len"][int(lrb[l]):int(lrb[l + 1])].sum()) for l in range(n)]
for l in np.argsort(sizes)[:3]:
    l = int(l)
    a0, a1 = int(lrb[l]), int(lrb[l + 1])
    reads = [bytes(b["read_blob"][int(b["read_off"][r]):int(b["read_off"][r]) + int(b["read_len"][r])]) for r in range(a0, a1)]
    lf = bytes(b["flank_blob"][int(b["lf_off"][l]):int(b["lf_off"][l]) + int(b["lf_len"][l])])
    rf = bytes(b["flank_blob"][int(b["rf_off"][l]):int(b["rf_off"][l]) + int(b["rf_len"][l])])
    tr = bytes(b["tr_blob"][int(b["tr_off"][l]):int(b["tr_off"][l]) + int(b["tr_len"][l])])
    m0, m1 = int(b["set_motif_begin"][l]), int(b["set_motif_begin"][l + 1])
    motifs = [bytes(b["motif_blob"][int(b["motif_off"][m]):int(b["motif_off"][m + 1])]) for m in range(m0, m1)]
    t0 = time.perf_counter(); ref = orc.locus_analyze(lf, rf, tr, motifs, reads); dt = time.perf_counter() - t0
    got = locus.locus_result(b, out, l).vcf_fields()
    ok = all(got[k] == ref[k] for k in ("AL", "ALLR", "SD", "MC", "MS", "AP")) and np.array_equal(out.span_start[a0:a1], ref["span_start"])
    print("locus", l, "read bytes", sizes[l], "oracle %.2f s" % dt, "parity", ok, got["AL"], got["MC"])
