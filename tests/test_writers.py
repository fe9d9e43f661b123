"""Native writers (trgt_amd/csrc/writers.hip) on a synthetic BAM with haplotype tags, 5mC calls and mismatching flank bases, fed with
hand-made genotyping results (no GPU): VCF lines incl. the AM field (write_vcf.rs:95-397; get_meth / assign_read / get_tr_meth,
tr.rs:196-262, 363-398) and the spanning-reads BAM (write_bam.rs:72-144; clip_bases.rs:9-120) against restatements in Python."""
import os

import pytest
import numpy as np

from bamtools import read_bam_records
from test_ingest import _synthetic


def _clip_cigar_by_bases(ref_pos, ops, left, right):
    """clip_bases.rs:59-118 restated on a per-base expansion of the CIGAR"""
    per_base = []  # one entry per query base: (op, reference position it sits at)
    r = ref_pos
    for c, n in ops:
        for _ in range(n):
            if c in "MIS=X":
                per_base.append((c, r))
            if c in "MDN=X":
                r += 1
    # reference position of the first kept base: every reference-consuming op in front of it has been consumed
    qi, r = 0, ref_pos
    out_pos = None
    kept = []
    total_q = len(per_base)
    for c, n in ops:
        for _ in range(n):
            if c in "MIS=X":
                if left <= qi < total_q - right:
                    if out_pos is None:
                        out_pos = r
                    kept.append(c)
                qi += 1
            elif left < qi < total_q - right or (left <= qi < total_q - right and kept):
                kept.append(c)  # a deletion between kept bases
            if c in "MDN=X":
                r += 1
    runs = []
    for c in kept:
        if runs and runs[-1][0] == c:
            runs[-1][1] += 1
        else:
            runs.append([c, 1])
    return out_pos, [(c, n) for c, n in runs]


def test_vcf_and_spanning_bam_from_handmade_results(tmp_path):
    from trgt_amd import ingest, locus, writers
    bam, fa, bed, recs, genome = _synthetic(tmp_path)
    rd = ingest.Reader(bam, fa)
    b = rd.batch(bed, keep_native=True)
    out = locus.BatchOutputs(b)
    nl, nr = b["n_loci"], b["n_reads"]
    names = b["read_name"]
    out.span_start[:] = -1
    out.span_end[:] = -1
    out.read_rank[:] = -1
    out.classification[:] = -1
    # locus 0: the two reads that span the region +- 500 get spans around the 60-base repeat; locus 1 stays without a call
    picks = {"spans_all": (0, 0, 500, 562), "rev_meth": (1, 1, 500, 560)}
    for r in range(nr):
        if names[r] in picks:
            rank, cls, s, e = picks[names[r]]
            out.span_start[r], out.span_end[r], out.read_rank[r], out.classification[r] = s, e, rank, cls
    seqs = {}
    for r in range(nr):
        if names[r] in picks:
            o = int(b["read_off"][r])
            seqs[picks[names[r]][0]] = bytes(b["read_blob"][o + picks[names[r]][2]:o + picks[names[r]][3]])
    out.n_alleles[0] = 2
    for a in (0, 1):
        o = int(out.allele_off[a])
        out.allele_blob[o:o + len(seqs[a])] = np.frombuffer(seqs[a], np.uint8)
        out.allele_len[a] = len(seqs[a])
        out.gt_size[a] = len(seqs[a])
        out.num_spanning[a] = 1
        out.purity[a] = 0.5 + 0.25 * a
        out.n_spans[a] = 1
        so = int(out.span_off[a])
        out.spans3[3 * so:3 * so + 3] = [a, 0, len(seqs[a])]
        co = int(out.count_off[a])
        out.motif_counts[co:co + 2] = [7 + a, 3]
    out.ci[:4] = [60, 60, 62, 62]
    out.ci[:4] = [min(len(seqs[0]), len(seqs[1])), max(len(seqs[0]), len(seqs[1]))] * 2
    w = writers.Writer(rd, tmp_path / "o.vcf", tmp_path / "o.bam", output_flank_len=40, sample_name="S1", command_line="cmd")
    w.write(b, out)
    w.close()
    lines = [l for l in open(tmp_path / "o.vcf").read().splitlines() if not l.startswith("##")]
    assert lines[0].endswith("\tS1") and len(lines) == 3
    f = lines[1].split("\t")
    pad = genome[1999]
    assert f[:2] == ["chr1", "2000"] and f[3] == pad + genome[2000:2060]
    alts = [pad + seqs[a].decode() for a in (0, 1) if seqs[a].decode() != genome[2000:2060]]
    assert f[4] == ",".join(dict.fromkeys(alts)) and f[7] == "TRID=L1;END=2060;MOTIFS=CAG,CCG;STRUC=(CAG)n(CCG)n"
    s = f[9].split(":")
    assert s[1] == "%d,%d" % (len(seqs[0]), len(seqs[1])) and s[3] == "1,1" and s[4] == "7_3,8_3" and s[6] == "0.500000,0.750000"
    assert s[5] == "0(0-%d),1(0-%d)" % (len(seqs[0]), len(seqs[1]))
    # AM: mean over the CpGs inside the span of (call / 255), per read; each read goes to the allele nearest in length whose interval holds it
    am = []
    by_name = {r["name"]: r for r in recs}
    for name in ("spans_all", "rev_meth"):
        r = list(names).index(name)
        o, n = int(b["read_off"][r]), int(b["read_len"][r])
        bases = bytes(b["read_blob"][o:o + n]).decode()
        me = list(b["meth"][int(b["meth_off"][r]):int(b["meth_off"][r + 1])])
        cpg = [i for i in range(n - 1) if bases[i:i + 2] == "CG"]
        assert len(cpg) == len(me)
        inside = [m / 255.0 for i, m in zip(cpg, me) if picks[name][2] <= i < picks[name][3]]
        am.append(None if not inside else sum(inside) / len(inside))
    lens = [len(seqs[0]), len(seqs[1])]
    exp = ["." if v is None else "%.2f" % v for v in am] if lens[0] != lens[1] else None
    if exp is not None:
        assert s[7] == ",".join(exp)
    assert lines[2].split("\t")[9] == ".:.:.:.:.:.:.:." and lines[2].split("\t")[4] == "."
    # ---- the spanning reads
    text, refs, got = read_bam_records(str(tmp_path / "o.bam"))
    assert refs == [("chr1", 6000), ("chr2", 6000)] and text.rstrip("\n").split("\n")[-1] == "@PG\tID:trgt\tPN:trgt\tCL:cmd\tVN:3.0.0"
    assert [g["name"] for g in got] == ["spans_all", "rev_meth"]
    for g in got:
        r = list(names).index(g["name"])
        rank, cls, s0, e0 = picks[g["name"]]
        o, n = int(b["read_off"][r]), int(b["read_len"][r])
        bases = bytes(b["read_blob"][o:o + n]).decode()
        left, right = s0 - 40, n - e0 - 40
        assert g["seq"] == bases[left:n - right] and g["qual"] == list(b["qual_blob"][o + left:o + n - right])
        cig = [("MIDNSHP=X"[int(v) & 0xF], int(v) >> 4) for v in b["cigar"][int(b["cigar_off"][r]):int(b["cigar_off"][r + 1])]]
        pos, ops = _clip_cigar_by_bases(int(b["cigar_ref_pos"][r]), cig, left, right)
        assert g["pos"] == pos and g["cigar"] == ops, g["name"]
        assert g["tid"] == 0 and g["mapq"] == 60 and g["flag"] == (20 if by_name[g["name"]]["flag"] & 16 else 4) and g["mtid"] == -1
        t = g["tags"]
        assert t["TR"] == ("Z", "L1") and t["AL"] == ("i", cls) and t["FL"] == ("BI", [40, 40])
        assert t["SO"] == ("i", int(b["start_offset"][r])) and t["EO"] == ("i", int(b["end_offset"][r]))
        assert t["HP"] == ("C", int(b["hp_tag"][r])) and abs(t["rq"][1] - 0.999) < 1e-6
        assert t["MO"] == ("Bi", list(b["mismatch_offsets"][int(b["mismatch_off"][r]):int(b["mismatch_off"][r + 1])]))
        me = list(b["meth"][int(b["meth_off"][r]):int(b["meth_off"][r + 1])])
        cpg = [i for i in range(n - 1) if bases[i:i + 2] == "CG"]
        assert t["MC"] == ("BC", [m for i, m in zip(cpg, me) if left <= i < n - right])
    # (the default keeps 0x4, as rust-htslib 0.46's Record::new() leaves it: UNPINNED, see the header) keep_unmapped_flag = 0 clears it
    w = writers.Writer(rd, tmp_path / "u.vcf", tmp_path / "u.bam", output_flank_len=40, sample_name="S1", command_line="cmd", keep_unmapped_flag=0)
    w.write(b, out)
    w.close()
    _, _, got_u = read_bam_records(str(tmp_path / "u.bam"))
    assert [(g["name"], g["flag"]) for g in got_u] == [(g["name"], g["flag"] & ~4) for g in got]
    assert [{k: v for k, v in g.items() if k != "flag"} for g in got_u] == [{k: v for k, v in g.items() if k != "flag"} for g in got]
    # ---- write_behind (ABI 11): the batch is formatted in the call, deflated and written behind it; the files are the same byte for byte
    # (three batches, so that writes wait for the batch before them; both compression levels of the BAM)
    for level in (6, 1):
        for wb in (0, 1):
            w = writers.Writer(rd, tmp_path / ("wb%d_%d.vcf" % (wb, level)), tmp_path / ("wb%d_%d.bam" % (wb, level)), output_flank_len=40, sample_name="S1",
                               command_line="cmd", bam_compress_level=level, write_behind=wb, threads=3)
            for _ in range(3):
                w.write(b, out)
            w.close()
        for ext in ("vcf", "bam"):
            assert open(tmp_path / ("wb0_%d.%s" % (level, ext)), "rb").read() == open(tmp_path / ("wb1_%d.%s" % (level, ext)), "rb").read(), (level, ext)
    assert len([l for l in open(tmp_path / "wb1_6.vcf").read().splitlines() if not l.startswith("#")]) == 6


def test_write_behind_reports_a_failed_write_at_the_next_call_or_the_close(tmp_path):
    """trgt_writer_params.write_behind: what the writer's own thread runs into (here: /dev/full, every flush fails) comes back from the
    next trgt_writer_write or from trgt_writer_close, never silently"""
    from trgt_amd import _lib, ingest, locus, writers
    if not os.path.exists("/dev/full"):
        pytest.skip("no /dev/full")
    bam, fa, bed, recs, genome = _synthetic(tmp_path)
    rd = ingest.Reader(bam, fa)
    b = rd.batch(bed, keep_native=True)
    out = locus.BatchOutputs(b)
    out.span_start[:] = -1; out.span_end[:] = -1; out.read_rank[:] = -1; out.classification[:] = -1
    for wb in (0, 1):
        w = writers.Writer(rd, "/dev/full", None, write_behind=wb)
        failed = 0
        for _ in range(3):  # (plain VCF through stdio: the buffer reaches the device at a flush or at the close)
            try:
                w.write(b, out)
            except _lib.TrgtHipError:
                failed += 1
        try:
            w.close()
        except _lib.TrgtHipError:
            failed += 1
        assert failed >= 1, wb
