"""Whole-catalog parity sweep: every locus of a synthetic catalog through trgt_locus_batch (reads resident in HBM) and through the
CPU oracle (all host cores), compared field by field as text records -- spans of every read, allele sequences, kept reads and their
classification, ALLR, SD, MC, MS, AP.  A developer tool (minutes of host time for 10^5..10^6 loci), not part of the test suite:
the suite holds the full-size batch to a seeded sample plus size-independent properties (tests/test_full_size_gpu.py).

    python tests/tools/parity_sweep.py <config> <n_loci> [first_locus] [chunk] [--host-reads | --bam4] [--host-glue] [--rq Q] [--depth D]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from trgt_amd import locus, synth, _lib
from oracle import binding as oracle


def gpu_records(b, out):
    nl = int(b["n_loci"])
    lrb = b["locus_read_begin"]
    recs = []
    for l in range(nl):
        a0, a1 = int(lrb[l]), int(lrb[l + 1])
        r = locus.locus_result(b, out, l)
        s = "S:" + "".join("%d,%d;" % (int(x), int(y)) for x, y in zip(out.span_start[a0:a1], out.span_end[a0:a1]))
        s += "|A:" + ",".join(a.seq.decode() for a in r.genotype)
        s += "|K:" + "".join("%d," % i for i in r.reads)
        s += "|C:" + "".join("%d," % c for c in r.classification)
        s += "|ALLR:" + "".join("%d-%d," % a.ci for a in r.genotype)
        s += "|SD:" + "".join("%d," % a.num_spanning for a in r.genotype)
        if r.genotype:
            f = r.vcf_fields()
            s += "|MC:" + f["MC"] + "|MS:" + f["MS"] + "|AP:" + f["AP"]
        recs.append(s)
    return recs


def main():
    argv = list(sys.argv[1:])
    host_reads = "--host-reads" in argv
    host_glue = "--host-glue" in argv
    bam4 = "--bam4" in argv  # reads handed over as BAM 4-bit codes (TRGT_READS_BAM4), from host memory
    argv = [a for a in argv if a not in ("--host-reads", "--host-glue", "--bam4")]
    rq_min, depth = None, None
    if "--rq" in argv:
        i = argv.index("--rq"); rq_min = float(argv[i + 1]); del argv[i:i + 2]
    if "--depth" in argv:
        i = argv.index("--depth"); depth = int(argv[i + 1]); del argv[i:i + 2]
    config, n_total = int(argv[0]), int(argv[1])
    first = int(argv[2]) if len(argv) > 2 else 0
    chunk = int(argv[3]) if len(argv) > 3 else 10000
    if host_glue:
        os.environ["TRGT_HOST_GENOTYPER"] = "1"
    params = locus.Params()
    okw = {}
    if rq_min is not None:
        params.min_read_qual = rq_min; okw["min_read_qual"] = rq_min
    if depth is not None:
        params.max_depth = depth; okw["max_depth"] = depth
    threads = min(os.cpu_count() or 1, 128)
    ctx = _lib.Context(0)
    bad = n_done = n_alleles = 0
    t_gpu = t_cpu = 0.0
    for c0 in range(first, first + n_total, chunk):
        n = min(chunk, first + n_total - c0)
        b = synth.generate_cfg3(n, first_locus=c0) if config == 3 else synth.generate(n, first_locus=c0, config=config)
        if rq_min is not None:  # rq tags: mostly high, some below the threshold, some missing
            rng = np.random.default_rng(c0 + 17)
            q = np.where(rng.random(int(b["n_reads"])) < 0.7, 0.99, 0.80 + 0.2 * rng.random(int(b["n_reads"])))
            q[rng.random(int(b["n_reads"])) < 0.05] = np.nan
            b["read_qual"] = np.ascontiguousarray(q, np.float64)
        rd, fd = (None, None) if host_reads or bam4 else (torch.from_numpy(b["read_blob"]).cuda(), torch.from_numpy(b["flank_blob"]).cuda())
        gb = locus.pack_bam4(b) if bam4 else b
        t0 = time.perf_counter()
        out = locus.run_batch(gb, params, ctx, flank_dev=fd, reads_dev=rd)
        t_gpu += time.perf_counter() - t0
        got = gpu_records(b, out)
        t0 = time.perf_counter()
        ref = oracle.locus_records(b, 0, n, threads, **okw)
        t_cpu += time.perf_counter() - t0
        for l, (g, r) in enumerate(zip(got, ref)):
            if g != r:
                bad += 1
                if bad <= 5:
                    print("MISMATCH locus %d\n  gpu    %s\n  oracle %s" % (c0 + l, g[:600], r[:600]))
        n_done += n
        n_alleles += int(out.n_alleles.sum())
        print("[sweep] config %d loci %d..%d: %d mismatches so far (gpu %.2f s, oracle %.1f s on %d threads)" % (config, first, c0 + n, bad, t_gpu, t_cpu, threads), flush=True)
    print("RESULT config=%d loci=%d alleles=%d mismatches=%d%s%s%s%s%s" % (config, n_done, n_alleles, bad, " bam4" if bam4 else "", " host-reads" if host_reads else "",
          " host-glue" if host_glue else "", " min_read_qual=%g" % rq_min if rq_min is not None else "", " max_depth=%d" % depth if depth is not None else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
