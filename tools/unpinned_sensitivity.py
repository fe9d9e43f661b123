#!/usr/bin/env python3
"""How many loci depend on the decisions no reference data pins?

The WFA arithmetic of the reference lives in WFA2-lib and its Ward linkage in kodama -- both un-vendored, so the oracle and the product
restate them from their published definitions.  Inside those restatements a few choices are recalled rather than verified.  This tool
re-runs synthetic catalogs of configs 2 / 4 / 5 through trgt_locus_batch with each choice flipped (developer switches TRGT_SENS_* of the `make DEV=1` library, read
when a context is created) and reports the fraction of loci whose alleles / read assignment / VCF fields change against the default run:

  bialign_min_length 0     WF_BIALIGN_FALLBACK_MIN_LENGTH read as 0 (SURVEY A.7 literally) instead of 100: short sequences go through the
                           breakpoint recursion instead of the unidirectional base case (wfaligner.rs:1754-1792 records both behaviours)
  consensus unidirectional the consensus alignments back-traced by plain WFA (MemoryHigh) instead of BiWFA: another co-optimal CIGAR
  ward ties last           nearest-neighbour ties of the NN-chain go to the last candidate instead of the first (kodama's order is recalled)
  lance-williams order     the update summed in another order: the last bits of the matrix central_read reads back

The reservoir's random stream (rand 0.9 StdRng) is not exercised: it only acts on loci with more reads than the reservoir holds, which
the synthetic catalogs (30 reads per locus) never have.

    python tools/unpinned_sensitivity.py [scale]      # scale 1.0: 100 k loci of cfg2 and cfg4, 20 k of cfg5
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the switches exist only in the developer build (make DEV=1 -> trgt_amd/libtrgt_hip_dev.so): use it, building it when it is missing
# (hipcc cross-compiles; on the GPU box the prebuilt file travels with the snapshot)
_DEV_SO = os.path.join(ROOT, "trgt_amd", "libtrgt_hip_dev.so")
os.environ.setdefault("TRGT_HIP_LIB", _DEV_SO)
import torch
from trgt_amd import locus, synth, _lib
if not os.path.exists(_DEV_SO):
    _lib.build_extension(dev=True)

VARIANTS = [
    ("bialign_min_length 0", dict(TRGT_SENS_BIALIGN_MIN_LEN=0)),
    ("consensus unidirectional", dict(TRGT_SENS_CONS_UNIDIR=1)),
    ("ward ties last", dict(TRGT_SENS_WARD_TIES=1)),
    ("lance-williams order", dict(TRGT_SENS_LW_ORDER=1)),
]


def records(b, out):
    """per locus: (allele sequences, everything else that reaches the VCF / BAM)"""
    recs = []
    for l in range(int(b["n_loci"])):
        r = locus.locus_result(b, out, l)
        al = tuple(a.seq for a in r.genotype)
        rest = (tuple(r.reads), tuple(r.classification), tuple(a.ci for a in r.genotype), tuple(a.num_spanning for a in r.genotype))
        if r.genotype:
            f = r.vcf_fields()
            rest += (f["MC"], f["MS"], f["AP"])
        recs.append((al, rest))
    return recs


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    plan = [(2, int(100000 * scale), 10000), (4, int(100000 * scale), 10000), (5, int(20000 * scale), 2000)]
    ctx0 = _lib.Context(0)
    ctxs = [(name, _lib.context_with_env(**env)) for name, env in VARIANTS]
    print("config  loci      variant                    alleles differ   any field differs   cluster loci")
    for cfg, n_total, chunk in plan:
        diff_al = {name: 0 for name, _ in VARIANTS}
        diff_any = {name: 0 for name, _ in VARIANTS}
        n_cluster = 0
        t0 = time.time()
        for first in range(0, n_total, chunk):
            n = min(chunk, n_total - first)
            b = synth.generate(n, first_locus=3_000_000 + first, config=cfg)
            if b.get("genotyper") is not None:
                n_cluster += int((b["genotyper"] == 1).sum())
            rd = torch.from_numpy(b["read_blob"]).cuda()
            fd = torch.from_numpy(b["flank_blob"]).cuda()
            base = records(b, locus.run_batch(b, locus.Params(), ctx0, flank_dev=fd, reads_dev=rd))
            for name, ctx in ctxs:
                got = records(b, locus.run_batch(b, locus.Params(), ctx, flank_dev=fd, reads_dev=rd))
                for x, y in zip(base, got):
                    diff_al[name] += x[0] != y[0]
                    diff_any[name] += x != y
        for name, _ in VARIANTS:
            print("cfg%d    %-8d  %-26s %7d (%.2e)   %7d (%.2e)   %d" % (cfg, n_total, name, diff_al[name], diff_al[name] / max(1, n_total),
                                                                        diff_any[name], diff_any[name] / max(1, n_total), n_cluster), flush=True)
        print("        (%.0f s)" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
