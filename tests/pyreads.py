"""Host-side mirror of the steps of analyze_tr that sit in FRONT of the GPU path: catalog + reference + BAM -> Locus + clipped reads.

Reference (PacificBiosciences/trgt v3.0.0):
  Locus (flanks from the genome)      src/trgt/locus.rs:13-23, 168-190        -> read_catalog
  extract_reads                       src/trgt/workflows/tr.rs:262-361        -> extract_reads
  HiFiRead::clip_to_region            src/trgt/reads/clip_region.rs:19-184    -> clip_to_region
  clip_reads                          src/trgt/workflows/tr.rs:186-196        -> clip_reads

Plain Python over zlib (BGZF is a multi-member gzip stream): the reference does this through htslib, which is I/O plumbing outside
the hot path; nothing here touches the GPU.  Not mirrored: methylation tags (MM/ML), SNV mismatch offsets, HP tags -- the locus path
behind them (get_meth, genotype_flank) is out of scope, so reads carry bases and the rq tag only; reservoir sampling of
loci deeper than 3 x max_depth (tr.rs:311-335) needs Rust's StdRng stream and raises NotImplementedError.
"""
import gzip
import struct
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

REF_CONSUMING = {0, 2, 3, 7, 8}  # M D N = X
QRY_CONSUMING = {0, 1, 4, 7, 8}  # M I S = X
SPLITTABLE = {0, 2, 3, 7, 8}
_SEQ_CODE = "=ACMGRSVTWYHKDBN"


@dataclass
class Locus:  # locus.rs:13-23
    id: str
    contig: str
    start: int
    end: int
    left_flank: bytes
    tr: bytes
    right_flank: bytes
    motifs: List[str]
    struc: str
    ploidy: int = 2
    genotyper: str = "size"


@dataclass
class BamRecord:
    name: str
    contig: Optional[str]
    pos: int
    flag: int
    cigar: List[Tuple[int, int]]  # (op code, length)
    seq: str
    rq: Optional[float]


def read_fasta(path) -> Dict[str, str]:
    seqs, name = {}, None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                name = line[1:].split()[0]
                seqs[name] = []
            elif name is not None:
                seqs[name].append(line)
    return {k: "".join(v) for k, v in seqs.items()}


def read_catalog(bed_path, genome: Dict[str, str], flank_len=250, genotyper="size") -> List[Locus]:
    """repeat catalog (BED: contig, start, end, ID=..;MOTIFS=..;STRUC=..) -> loci with upper-cased flanks (locus.rs:168-190)"""
    loci = []
    for line in open(bed_path):
        if not line.strip():
            continue
        contig, start, end, info = line.split()[:4]
        start, end = int(start), int(end)
        f = dict(x.split("=", 1) for x in info.split(";"))
        g = genome[contig]
        if start < flank_len or end + flank_len > len(g):
            raise ValueError("locus %s: flanks leave the contig" % f["ID"])
        loci.append(Locus(f["ID"], contig, start, end, g[start - flank_len:start].upper().encode(), g[start:end].upper().encode(),
                          g[end:end + flank_len].upper().encode(), f["MOTIFS"].split(","), f["STRUC"], 2, genotyper))
    return loci


def _rq_tag(buf):
    i = 0
    size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    while i < len(buf):
        tag, ty = buf[i:i + 2], chr(buf[i + 2])
        i += 3
        if ty in size:
            if tag == b"rq" and ty == "f":
                return struct.unpack_from("<f", buf, i)[0]
            i += size[ty]
        elif ty in "ZH":
            i = buf.index(b"\0", i) + 1
        elif ty == "B":
            sub, n = chr(buf[i]), struct.unpack_from("<I", buf, i + 1)[0]
            i += 5 + n * size[sub]
        else:
            raise ValueError("bad BAM tag type " + ty)
    return None


def read_bam(path) -> List[BamRecord]:
    data = gzip.open(path, "rb").read()
    if data[:4] != b"BAM\1":
        raise ValueError("not a BAM file")
    p = 8 + struct.unpack_from("<i", data, 4)[0]
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, p)[0]
        refs.append(data[p + 4:p + 4 + l_name - 1].decode())
        p += 8 + l_name
    out = []
    while p < len(data):
        bs = struct.unpack_from("<i", data, p)[0]
        rec = data[p + 4:p + 4 + bs]
        p += 4 + bs
        ref_id, pos, l_rn, _mapq, _bin, n_cig, flag, l_seq, _nr, _np, _tl = struct.unpack_from("<iiBBHHHiiii", rec, 0)
        q = 32
        name = rec[q:q + l_rn - 1].decode()
        q += l_rn
        cigar = [(v & 0xF, v >> 4) for v in struct.unpack_from("<%dI" % n_cig, rec, q)]
        q += 4 * n_cig
        packed = rec[q:q + (l_seq + 1) // 2]
        q += (l_seq + 1) // 2 + l_seq
        seq = "".join(_SEQ_CODE[(packed[i >> 1] >> (4 if i % 2 == 0 else 0)) & 0xF] for i in range(l_seq))
        out.append(BamRecord(name, refs[ref_id] if ref_id >= 0 else None, pos, flag, cigar, seq, _rq_tag(rec[q:])))
    return out


def extract_reads(locus: Locus, records: List[BamRecord], flank_len=250, min_read_qual=0.98, max_depth=250):
    """tr.rs:262-361: records overlapping region +- flank_len, in file order; returns (records, number dropped for quality)"""
    lo, hi = max(0, locus.start - flank_len), locus.end + flank_len
    reads, n_filt = [], 0
    for r in records:
        if r.contig != locus.contig or (r.flag & 0x4):
            continue
        ref_end = r.pos + sum(n for c, n in r.cigar if c in REF_CONSUMING)
        if ref_end <= lo or hi <= r.pos:
            continue
        if r.flag & (0x100 | 0x800):  # secondary / supplementary
            continue
        if (r.rq if r.rq is not None else 1.0) < min_read_qual:
            n_filt += 1
            continue
        reads.append(r)
    if len(reads) >= 3 * max_depth:
        raise NotImplementedError("reservoir sampling of deep loci (tr.rs:311-335) is not mirrored")
    return reads, n_filt


def clip_cigar(ref_pos0, ops, region):
    """clip_region.rs:108-184: (ref_start, query_start, ops) of the part of the alignment inside region, or None"""
    rs, re_ = region
    rlen = lambda o: o[1] if o[0] in REF_CONSUMING else 0
    qlen = lambda o: o[1] if o[0] in QRY_CONSUMING else 0
    if ref_pos0 + sum(rlen(o) for o in ops) <= rs or re_ <= ref_pos0:
        return None
    ref_pos, query_pos, i, out = ref_pos0, 0, 0, []
    while i < len(ops) and ref_pos + rlen(ops[i]) <= rs:
        ref_pos += rlen(ops[i]); query_pos += qlen(ops[i]); i += 1
    c_ref, c_qry = ref_pos, query_pos
    if ref_pos < rs:
        outside, op = rs - ref_pos, ops[i]
        assert op[0] in SPLITTABLE
        out.append((op[0], rlen(op) - outside if ref_pos + rlen(op) <= re_ else re_ - rs))
        c_ref += outside
        if qlen(out[-1]) != 0:
            c_qry += outside
        ref_pos += rlen(op); query_pos += qlen(op); i += 1
    while i < len(ops) and ref_pos + rlen(ops[i]) <= re_:
        out.append(ops[i]); ref_pos += rlen(ops[i]); query_pos += qlen(ops[i]); i += 1
    if i < len(ops) and ref_pos < re_:
        assert ops[i][0] in SPLITTABLE
        out.append((ops[i][0], re_ - ref_pos))
    return c_ref, c_qry, out


def clip_to_region(rec: BamRecord, region) -> Optional[bytes]:
    """HiFiRead::clip_to_region (clip_region.rs:19-76), bases only"""
    r = clip_cigar(rec.pos, rec.cigar, region)
    if r is None:
        return None
    _, q0, ops = r
    n = sum(o[1] for o in ops if o[0] in QRY_CONSUMING)
    return rec.seq[q0:q0 + n].encode()


def clip_reads(locus: Locus, radius: int, reads: List[BamRecord]):
    """tr.rs:186-196; returns (clipped bases, rq) of the reads that overlap region +- radius"""
    region = (locus.start - radius, locus.end + radius)
    out = []
    for r in reads:
        s = clip_to_region(r, region)
        if s is not None:
            out.append((s, r.rq))
    return out


def locus_inputs(locus: Locus, records: List[BamRecord], flank_len=250, min_read_qual=0.98, max_depth=250):
    """Everything analyze_tr does before get_spanning_reads (tr.rs:29-35): the dict trgt_amd.locus.pack takes."""
    reads, _ = extract_reads(locus, records, flank_len, min_read_qual, max_depth)
    clipped = clip_reads(locus, 2 * flank_len, reads)
    return dict(left_flank=locus.left_flank, right_flank=locus.right_flank, tr=locus.tr, motifs=[m.encode() for m in locus.motifs],
                ploidy=locus.ploidy, genotyper=locus.genotyper, reads=[s for s, _ in clipped], read_qual=[q for _, q in clipped])
