"""Synthetic end-to-end data set: a genome (FASTA + .fai), a repeat catalog (BED) and a coordinate-sorted BAM (+ .bai) of full-length
reads over cfg2-like loci (motif 2-6 bp, 5-40 copies per allele, two alleles, 250 bp flanks), written with numpy +
zlib so that a data set of a few hundred MB takes seconds.  It feeds bench.py's end-to-end leg (BAM -> native ingestion -> GPU -> VCF /
spanning BAM writer); the small hand-made records of tests/bamtools.py stay the tool for the ingestion tests proper.

The reads are what a mapper would hand over: `read_len` bases of genome around the locus with the allele of the read's haplotype in the
place of the reference repeat (CIGAR <left>M <diff>I|D <right>M), substitutions at `sub_rate`, an rq tag, binned HiFi-like qualities (85 % in the top bin).
"""
import os
import struct
import zlib

import numpy as np

_BASES = np.frombuffer(b"ACGT", np.uint8)
_CODE = np.zeros(256, np.uint8)
for _i, _c in enumerate(b"=ACMGRSVTWYHKDBN"):
    _CODE[_c] = _i


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def _bgzf_block(data, level):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def write_dataset(dirpath, n_loci=1000, read_len=6000, reads_per_locus=30, seed=20250928, flank_len=250, sub_rate=1e-3, level=1):
    """Writes <dir>/genome.fa(.fai), <dir>/catalog.bed, <dir>/reads.bam(.bai); returns a dict with the paths and sizes."""
    rng = np.random.default_rng(seed)
    os.makedirs(dirpath, exist_ok=True)
    spacer = read_len + 200
    # ---- loci: motif, reference copy number, the two alleles ----------------------------------------------------------------------
    motifs, ref_tr, alleles = [], [], []
    for _ in range(n_loci):
        m = _BASES[rng.integers(0, 4, size=int(rng.integers(2, 7)))]
        while len(m) > 1 and (m == m[0]).all():
            m = _BASES[rng.integers(0, 4, size=len(m))]
        motifs.append(m)
        ref_tr.append(np.tile(m, int(rng.integers(5, 41))))
        alleles.append([np.tile(m, int(rng.integers(5, 41))) for _ in range(2)])
    parts, starts, pos = [], [], 0
    for l in range(n_loci):
        sp = _BASES[rng.integers(0, 4, size=spacer)]
        parts += [sp, ref_tr[l]]
        starts.append(pos + spacer)
        pos += spacer + len(ref_tr[l])
    parts.append(_BASES[rng.integers(0, 4, size=spacer)])
    genome = np.concatenate(parts)
    glen = len(genome)
    # ---- FASTA + .fai, catalog ---------------------------------------------------------------------------------------------------
    width = 60
    fa = os.path.join(dirpath, "genome.fa")
    full = (glen // width) * width
    body = np.empty((glen // width, width + 1), np.uint8)
    body[:, :width] = genome[:full].reshape(-1, width)
    body[:, width] = 10
    with open(fa, "wb") as f:
        f.write(b">chr1\n")
        f.write(body.tobytes())
        if glen > full:
            f.write(genome[full:].tobytes() + b"\n")
    open(fa + ".fai", "w").write("chr1\t%d\t%d\t%d\t%d\n" % (glen, 6, width, width + 1))
    bed = os.path.join(dirpath, "catalog.bed")
    with open(bed, "w") as f:
        for l in range(n_loci):
            m = motifs[l].tobytes().decode()
            f.write("chr1\t%d\t%d\tID=L%d;MOTIFS=%s;STRUC=(%s)n\n" % (starts[l], starts[l] + len(ref_tr[l]), l, m, m))
    # ---- records -------------------------------------------------------------------------------------------------------------------
    recs = []  # (pos, end, bytes)
    half = read_len // 2
    rq = struct.pack("<f", 0.999)
    for l in range(n_loci):
        s, e = starts[l], starts[l] + len(ref_tr[l])
        for i in range(reads_per_locus):
            al = alleles[l][i & 1]
            lc = half + int(rng.integers(-half // 4, half // 4))
            rc = max(flank_len + 50, read_len - lc - len(al))
            seq = np.concatenate([genome[s - lc:s], al, genome[e:e + rc]])
            n = len(seq)
            k = rng.binomial(n, sub_rate)
            if k:
                at = rng.integers(0, n, size=k)
                seq[at] = _BASES[rng.integers(0, 4, size=k)]
            d = len(al) - (e - s)
            common = min(len(al), e - s)
            cig = [((lc + common) << 4) | 0]
            if d > 0:
                cig.append((d << 4) | 1)
            elif d < 0:
                cig.append(((-d) << 4) | 2)
            cig.append((rc << 4) | 0)
            codes = _CODE[seq]
            if n & 1:
                codes = np.append(codes, np.uint8(0))
            packed = ((codes[0::2] << 4) | codes[1::2]).tobytes()
            name = b"m%d/%d/ccs\0" % (l, i)
            p0, p1 = s - lc, e + rc
            head = struct.pack("<iiBBHHHiiii", 0, p0, len(name), 60, _reg2bin(p0, p1), len(cig), 16 if (i & 2) else 0, n, -1, -1, 0)
            qual = np.full(n, 40, np.uint8)
            lowq = rng.random(n) < 0.15
            qual[lowq] = rng.integers(2, 40, size=int(lowq.sum()))
            rec = head + name + struct.pack("<%dI" % len(cig), *cig) + packed + qual.tobytes() + b"rqf" + rq
            recs.append((p0, p1, struct.pack("<i", len(rec)) + rec))
    recs.sort(key=lambda r: r[0])
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:%d\n" % glen
    headb = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", glen)
    ustart = np.zeros(len(recs) + 1, np.int64)
    ustart[0] = len(headb)
    ustart[1:] = len(headb) + np.cumsum([len(r[2]) for r in recs])
    stream = headb + b"".join(r[2] for r in recs)
    block = 0xff00
    bam = os.path.join(dirpath, "reads.bam")
    coffs, at = [], 0
    with open(bam, "wb") as f:
        for o in range(0, len(stream), block):
            coffs.append(at)
            b = _bgzf_block(stream[o:o + block], level)
            f.write(b)
            at += len(b)
        eof_coff = at
        f.write(_bgzf_block(b"", level))
    coffs.append(eof_coff)

    def voff(u):
        b = int(u) // block
        return (coffs[b] << 16) | (int(u) % block) if b < len(coffs) - 1 else eof_coff << 16

    bins, linear = {}, {}
    for k, (p0, p1, _) in enumerate(recs):
        v0, v1 = voff(ustart[k]), voff(ustart[k + 1])
        bins.setdefault(_reg2bin(p0, p1), []).append((v0, v1))
        for w in range(p0 >> 14, ((p1 - 1) >> 14) + 1):
            if w not in linear or v0 < linear[w]:
                linear[w] = v0
    idx = [b"BAI\1", struct.pack("<ii", 1, len(bins))]
    for b, chunks in sorted(bins.items()):
        merged = [list(chunks[0])]  # file order: neighbours coalesce
        for a, e2 in chunks[1:]:
            if a == merged[-1][1]:
                merged[-1][1] = e2
            else:
                merged.append([a, e2])
        idx.append(struct.pack("<Ii", b, len(merged)) + b"".join(struct.pack("<QQ", a, e2) for a, e2 in merged))
    n_intv = (max(linear) + 1) if linear else 0
    idx.append(struct.pack("<i", n_intv))
    last = 0
    for w in range(n_intv):
        last = linear.get(w, last)
        idx.append(struct.pack("<Q", last))
    open(bam + ".bai", "wb").write(b"".join(idx))
    return dict(bam=bam, fasta=fa, bed=bed, n_loci=n_loci, n_reads=len(recs), bam_bytes=os.path.getsize(bam), bases=int(sum(len(r[2]) for r in recs)),
                read_len=read_len, allele_len=np.array([[len(a[0]), len(a[1])] for a in alleles], np.int64))
