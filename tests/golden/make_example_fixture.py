#!/usr/bin/env python3
"""Build tests/golden/example_e1_reads.json from the reference's example data set.

Run in the build container only (needs /root/reference):

    python tests/golden/make_example_fixture.py

What it does -- the data-preparation half of `analyze_tr` that sits *before* the hot path, restated in Python so the
hot path can be fed exactly what the reference feeds it:

  * locus from example/repeat.bed + example/reference.fasta: flanks of --flank-len 250 and the TR
    (src/trgt/locus.rs:168-190: left = [start-250, start), tr = [start, end), right = [end, end+250), upper-cased)
  * reads from example/sample.bam (BGZF/BAM parsed here, no htslib): drop secondary/supplementary and rq < 0.98
    (src/trgt/workflows/tr.rs:268-305), keep records overlapping region +-250 in file order
  * clip every read to region +-2*250 on the reference axis (tr.rs:33-34, 186-196 and
    src/trgt/reads/clip_region.rs:19-76, 108-184), keeping the clipped bases

The expected outputs (AL/ALLR/SD/MC/MS/AP of the VCF record the reference prints for this input) are the tutorial's,
docs/tutorial.md:29-46, already held in example_e1.json.
"""
import gzip
import json
import os
import struct
import sys

REF = "/root/reference/example"
HERE = os.path.dirname(os.path.abspath(__file__))
FLANK_LEN = 250
MIN_RQ = 0.98
MAX_DEPTH = 250

REF_CONSUMING = {0, 2, 3, 7, 8}  # M D N = X
QRY_CONSUMING = {0, 1, 4, 7, 8}  # M I S = X
SPLITTABLE = {0, 2, 3, 7, 8}
SEQ_CODE = "=ACMGRSVTWYHKDBN"


def read_fasta(path):
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


def parse_tags(buf):
    tags, i = {}, 0
    sizes = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    fmts = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    while i < len(buf):
        tag = buf[i:i + 2].decode()
        ty = chr(buf[i + 2])
        i += 3
        if ty == "A":
            tags[tag] = chr(buf[i]); i += 1
        elif ty in fmts:
            tags[tag] = struct.unpack_from(fmts[ty], buf, i)[0]; i += sizes[ty]
        elif ty in "ZH":
            j = buf.index(b"\0", i)
            tags[tag] = buf[i:j].decode(); i = j + 1
        elif ty == "B":
            sub = chr(buf[i]); n = struct.unpack_from("<I", buf, i + 1)[0]
            i += 5 + n * sizes[sub]
            tags[tag] = None
        else:
            raise ValueError("bad tag type " + ty)
    return tags


def read_bam(path):
    data = gzip.open(path, "rb").read()  # BGZF is a multi-member gzip stream
    assert data[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", data, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]; p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, p)[0]; p += 4
        refs.append(data[p:p + l_name - 1].decode()); p += l_name + 4
    recs = []
    while p < len(data):
        bs = struct.unpack_from("<i", data, p)[0]; p += 4
        rec = data[p:p + bs]; p += bs
        ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, _nr, _np, _tl = struct.unpack_from("<iiBBHHHiiii", rec, 0)
        q = 32
        name = rec[q:q + l_rn - 1].decode(); q += l_rn
        cigar = [(v & 0xF, v >> 4) for v in struct.unpack_from("<%dI" % n_cig, rec, q)]; q += 4 * n_cig
        packed = rec[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        seq = "".join(SEQ_CODE[(packed[i >> 1] >> (4 if i % 2 == 0 else 0)) & 0xF] for i in range(l_seq))
        q += l_seq
        tags = parse_tags(rec[q:])
        recs.append(dict(name=name, contig=refs[ref_id] if ref_id >= 0 else None, pos=pos, flag=flag, mapq=mapq,
                         cigar=cigar, seq=seq, tags=tags))
    return recs


def clip_cigar(ref_pos0, ops, region):
    """clip_region.rs:108-184. ops = [(code, len)], returns (ref_start, query_start, ops) or None."""
    rs, re_ = region
    rlen = lambda o: o[1] if o[0] in REF_CONSUMING else 0
    qlen = lambda o: o[1] if o[0] in QRY_CONSUMING else 0
    read_end = ref_pos0 + sum(rlen(o) for o in ops)
    if read_end <= rs or re_ <= ref_pos0:
        return None
    ref_pos, query_pos, i, out = ref_pos0, 0, 0, []
    while i < len(ops) and ref_pos + rlen(ops[i]) <= rs:
        ref_pos += rlen(ops[i]); query_pos += qlen(ops[i]); i += 1
    c_ref, c_qry = ref_pos, query_pos
    if ref_pos < rs:
        outside = rs - ref_pos
        op = ops[i]
        assert op[0] in SPLITTABLE
        n = rlen(op) - outside if ref_pos + rlen(op) <= re_ else re_ - rs
        out.append((op[0], n))
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


def clip_to_region(rec, region):
    r = clip_cigar(rec["pos"], rec["cigar"], region)
    if r is None:
        return None
    _, q0, ops = r
    n = sum(o[1] for o in ops if o[0] in QRY_CONSUMING)
    return rec["seq"][q0:q0 + n]


def main():
    genome = read_fasta(os.path.join(REF, "reference.fasta"))
    loci = []
    for line in open(os.path.join(REF, "repeat.bed")):
        contig, start, end, info = line.split()
        fields = dict(f.split("=", 1) for f in info.split(";"))
        loci.append((contig, int(start), int(end), fields))
    recs = read_bam(os.path.join(REF, "sample.bam"))
    out = []
    for contig, start, end, fields in loci:
        g = genome[contig]
        lf, tr, rf = g[start - FLANK_LEN:start].upper(), g[start:end].upper(), g[end:end + FLANK_LEN].upper()
        fetch = (max(0, start - FLANK_LEN), end + FLANK_LEN)
        reads, n_filt = [], 0
        for r in recs:
            if r["contig"] != contig or (r["flag"] & 0x4):
                continue
            ref_end = r["pos"] + sum(n for c, n in r["cigar"] if c in REF_CONSUMING)
            if ref_end <= fetch[0] or fetch[1] <= r["pos"]:
                continue
            if r["flag"] & (0x100 | 0x800):
                continue
            rq = r["tags"].get("rq")
            if (rq if rq is not None else 1.0) < MIN_RQ:
                n_filt += 1
                continue
            reads.append(r)
        assert len(reads) < 3 * MAX_DEPTH, "reservoir sampling would kick in; not restated here"
        region = (start - 2 * FLANK_LEN, end + 2 * FLANK_LEN)
        clipped = [s for s in (clip_to_region(r, region) for r in reads) if s is not None]
        out.append(dict(id=fields["ID"], motifs=fields["MOTIFS"].split(","), struc=fields["STRUC"], contig=contig,
                        start=start, end=end, left_flank=lf, tr=tr, right_flank=rf, reads=clipped,
                        n_quality_filtered=n_filt))
        print(f"{fields['ID']}: {len(clipped)} reads (filtered {n_filt}), tr {len(tr)} bp", file=sys.stderr)
    with open(os.path.join(HERE, "example_e1_reads.json"), "w") as f:
        json.dump(dict(src="example/{repeat.bed,reference.fasta,sample.bam}; genotype defaults "
                           "(--flank-len 250, min rq 0.98, --max-depth 250)", loci=out), f, indent=1)


if __name__ == "__main__":
    main()
