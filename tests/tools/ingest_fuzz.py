"""Fuzz of the device-side read ingestion (trgt_ingest_params.ingest_device) against the host path of the same library: random BAM files
written by tests/bamtools.py -- reads with random CIGARs (M = X I D S N H P runs, soft clips at the ends), both strands, secondary /
supplementary / unmapped flags, rq values either side of the filter, HP tags, MM / ML tags of several shapes (C+m with and without '?',
numeric codes, two codes, other bases, deltas that run off the read, too few ML values), read lengths from 30 bases to several kb, loci
with no read and loci deeper than the reservoir, several contigs, catalog windows cut by flank_len 20 .. 250 -- every array of the two
batches compared.  Usage: python tests/tools/ingest_fuzz.py [n_files] [seed]"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bamtools import write_bam, write_fasta  # noqa: E402
from trgt_amd import ingest  # noqa: E402


def comp(s):
    return s.translate(str.maketrans("ACGT", "TGCA"))[::-1]


def one_file(rng, d, tag):
    glen = int(rng.integers(6000, 30000))
    contigs = [("chr1", "".join(rng.choice(list("ACGT"), glen))), ("chr2", "".join(rng.choice(list("ACGT"), 4000)))]
    fa = os.path.join(d, tag + ".fa")
    write_fasta(fa, contigs)
    flank = int(rng.choice([20, 50, 100, 250]))
    n_loci = int(rng.integers(1, 8))
    loci, at = [], 2 * flank + 300
    for l in range(n_loci):
        w = int(rng.integers(6, 90))
        if at + w + 2 * flank + 10 >= glen:
            break
        loci.append((at, at + w))
        at += w + int(rng.integers(50, 1500))
    bed = os.path.join(d, tag + ".bed")
    with open(bed, "w") as f:
        for i, (a, b) in enumerate(loci):
            f.write("chr1\t%d\t%d\tID=F%d;MOTIFS=CAG;STRUC=(CAG)n\n" % (a, b, i))
        f.write("chr2\t1000\t1030\tID=other;MOTIFS=A;STRUC=(A)n\n")
    recs = []
    deep = int(rng.integers(0, len(loci))) if rng.random() < 0.3 else -1
    for li, (a, b) in enumerate(loci):
        n_reads = int(rng.integers(0, 40)) if li != deep else int(rng.integers(100, 200))
        for r in range(n_reads):
            span = int(rng.integers(30, 4000))
            pos = max(0, a - int(rng.integers(0, span)))
            ops, ref, left = [], pos, span
            if rng.random() < 0.3:
                ops.append(("S", int(rng.integers(1, 20))))
            if rng.random() < 0.05:
                ops.insert(0, ("H", int(rng.integers(1, 9))))
            while left > 0:
                c = str(rng.choice(list("M=XIDN"), p=[0.25, 0.35, 0.15, 0.1, 0.1, 0.05]))
                n = int(rng.integers(1, 8)) if c in "XIDN" else int(rng.integers(1, min(left, 400) + 1))
                if ops and ops[-1][0] == c:
                    continue
                if c != "I" and ref + n >= glen:
                    break
                ops.append((c, n))
                if c != "I":
                    ref += n
                left -= n
            if not any(c in "M=X" for c, _ in ops):
                ops.append(("M", 5))
            if rng.random() < 0.3:
                ops.append(("S", int(rng.integers(1, 20))))
            qlen = sum(n for c, n in ops if c in "MIS=X")
            seq = "".join(rng.choice(list("ACGT"), qlen))
            if rng.random() < 0.5:  # CpG-rich stretch
                k = int(rng.integers(0, max(1, qlen - 40)))
                seq = seq[:k] + ("CG" * 20)[:max(0, min(40, qlen - k))] + seq[k + 40:]
                seq = seq[:qlen]
            flag = int(rng.choice([0, 16, 0, 16, 256, 2048, 4, 20], p=[0.4, 0.4, 0.05, 0.05, 0.03, 0.03, 0.02, 0.02]))
            tags = {}
            if rng.random() < 0.9:
                tags["rq"] = ("f", float(rng.choice([0.999, 0.99, 0.95, 0.5])))
            if rng.random() < 0.4:
                tags["HP"] = ("C", int(rng.integers(0, 3)))
            shape = int(rng.integers(0, 9))
            if shape < 7:
                orig = comp(seq) if flag & 16 else seq
                n_c = orig.count("C")
                nd = int(rng.integers(0, max(1, min(n_c, 60))))
                deltas = [int(rng.integers(0, 3)) for _ in range(nd)]
                if shape == 5 and deltas:
                    deltas[-1] = 100000
                head = ["C+m", "C+m?", "C+m.", "C+76792", "C+hm", "C+m", "C+m"][shape]
                mm = head + "".join(",%d" % x for x in deltas) + ";"
                if shape == 3:
                    mm += "C+m,0,0;"
                if shape == 6:
                    mm = "A+a,1;" + mm + "C+m,1;"
                n_ml = mm.count(",") * (2 if "hm" in mm else 1) - (2 if shape == 4 and nd > 2 else 0)
                tags["MM"] = ("Z", mm)
                tags["ML"] = ("BC", [int(x) for x in rng.integers(1, 255, size=max(0, n_ml))])
            recs.append(dict(name="r%d_%d" % (li, r), tid=0, pos=pos, cigar=ops, seq=seq, flag=flag, mapq=int(rng.integers(0, 61)), tags=tags,
                             qual=[int(x) for x in rng.integers(2, 60, qlen)]))
    recs.append(dict(name="c2", tid=1, pos=900, cigar=[("M", 300)], seq="".join(rng.choice(list("ACGT"), 300)), tags={"rq": ("f", 0.999)}))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    bam = os.path.join(d, tag + ".bam")
    write_bam(bam, [("chr1", glen), ("chr2", 4000)], recs, block=int(rng.choice([0x1000, 0x4000, 0x8000, 0xff00])))
    return bam, fa, bed, flank, len(recs)


def same(x, y):
    for k in x:
        if k in ("read_blob_dev", "read_blob_device", "_native"):
            continue
        if isinstance(x[k], np.ndarray):
            ok = x[k].dtype == y[k].dtype and x[k].shape == y[k].shape and (np.array_equal(x[k], y[k], equal_nan=True) if x[k].dtype.kind == "f" else np.array_equal(x[k], y[k]))
        else:
            ok = x[k] == y[k]
        if not ok:
            return k
    return None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = reads = fell = 0
    with tempfile.TemporaryDirectory() as d:
        for i in range(n):
            bam, fa, bed, flank, nrec = one_file(rng, d, "f%d" % i)
            rd = ingest.Reader(bam, fa)
            kw = dict(flank_len=flank, max_depth=int(rng.choice([10, 250])), min_read_qual=float(rng.choice([0.98, 0.9, 0.0])), keep_bam4=int(rng.integers(0, 2)))
            host = rd.batch(bed, **kw)
            dev = rd.batch(bed, ingest_device=0, **kw)
            st = rd.device_stats()
            fell += st["fallbacks"]
            k = same(host, dev)
            reads += int(host["n_reads"])
            if k is not None or host["n_reads"] != dev["n_reads"]:
                bad += 1
                extra = ""
                if k is not None and isinstance(host[k], np.ndarray) and host[k].shape == dev[k].shape:
                    neq = np.flatnonzero(~((host[k] == dev[k]) | ((host[k] != host[k]) & (dev[k] != dev[k]))))
                    extra = " first at %d of %d: host %r, device %r (%d differ); kw %r" % (int(neq[0]), len(host[k]), host[k][neq[0]], dev[k][neq[0]], len(neq), kw)
                print("MISMATCH file %d (seed %d): %s%s" % (i, seed, k, extra), flush=True)
            rd.close()
            for p in (bam, bam + ".bai", fa, fa + ".fai", bed):
                os.remove(p)
    print("RESULT ingest_fuzz: %d files, %d clipped reads, %d calls fell back to the host path, %d mismatches" % (n, reads, fell, bad))
    sys.exit(1 if bad else 0)


main()
