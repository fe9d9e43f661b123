#!/usr/bin/env python3
"""Generate tests/golden/{wfa,hmm}_kats.json.

Run ONCE in the build container (needs /root/reference, which does not exist on
the GPU box).  The fixtures are DATA: the input sequences and expected outputs
of the reference's own known-answer tests, each tagged
with the reference file:line it was transcribed from.  Long literals are pulled
from the reference sources by regex so they are not re-typed by hand; no
reference source text is stored, only the test vectors.
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def rs(path):
    return open(os.path.join(REF, path)).read()


def literal_after(src, anchor, name, nth=0):
    """First byte-string literal assigned to `let <name> = b"..."` after `anchor`."""
    i = src.index(anchor)
    pat = re.compile(r"let\s+%s\s*=\s*b\"([A-Z]+)\"" % re.escape(name))
    ms = list(pat.finditer(src, i))
    return ms[nth].group(1)


def main():
    wf = rs("src/wfaligner.rs")
    PATTERN = re.search(r'const PATTERN: &\[u8\] = b"([A-Z]+)"', wf).group(1)
    TEXT = re.search(r'const TEXT: &\[u8\] = b"([A-Z]+)"', wf).group(1)

    def P(**kw):
        d = dict(metric="affine", x=0, o1=0, e1=0, o2=0, e2=0, span="end2end", pbf=0, pef=0, tbf=0, tef=0,
                 scope="alignment", memory="high", heuristic="default")
        d.update(kw)
        return d

    kats = []
    # W1-W6  wfaligner.rs:1136-1227
    kats.append(dict(id="W1_indel", src="src/wfaligner.rs:1136-1150", params=P(metric="indel"), pattern=PATTERN, text=TEXT,
                     status=0, score=10, cigar="1M1I1D3M1I5M2I2D8M1I1M1I1M1I9M"))
    kats.append(dict(id="W2_edit", src="src/wfaligner.rs:1152-1166", params=P(metric="edit"), pattern=PATTERN, text=TEXT,
                     status=0, score=7, cigar="1M1X3M1I5M2X8M1I1M1I1M1I9M"))
    kats.append(dict(id="W3_linear", src="src/wfaligner.rs:1168-1182", params=P(metric="linear", x=6, e1=2), pattern=PATTERN,
                     text=TEXT, status=0, score=-20, cigar="1M1I1D3M1I5M2I2D8M1I1M1I1M1I9M"))
    kats.append(dict(id="W4_affine_lowmem", src="src/wfaligner.rs:1184-1198", params=P(x=6, o1=4, e1=2, memory="low"),
                     pattern=PATTERN, text=TEXT, status=0, score=-40, cigar="1M1X3M1I5M2X8M3I1M1X9M"))
    kats.append(dict(id="W5_affine_score_only", src="src/wfaligner.rs:1200-1211",
                     params=P(x=6, o1=4, e1=2, memory="low", scope="score"), pattern=PATTERN, text=TEXT, status=0, score=-40,
                     cigar=""))
    kats.append(dict(id="W6_affine2p", src="src/wfaligner.rs:1213-1227", params=P(metric="affine2p", x=6, o1=2, e1=2, o2=4, e2=1),
                     pattern=PATTERN, text=TEXT, status=0, score=-34, cigar="1M1X3M1I5M2X8M1I1M1I1M1I9M"))
    # W7 / W8 spans  wfaligner.rs:1229-1261
    p7, t7 = "AATTTAAGTCTAGGCTACTTTC", "CCGACTACTACGAAATTTAAGTATAGGCTACTTTCCGTACGTACGTACGT"
    kats.append(dict(id="W7_span_1", src="src/wfaligner.rs:1229-1243",
                     params=P(metric="affine2p", x=8, o1=4, e1=2, o2=24, e2=1, span="endsfree", tef=len(t7)), pattern=p7,
                     text=t7, status=0, span=[0, 22, 13, 35]))
    p8 = literal_after(wf, "fn test_aligner_span_2", "pattern")
    t8 = literal_after(wf, "fn test_aligner_span_2", "text")
    kats.append(dict(id="W8_span_2", src="src/wfaligner.rs:1245-1261",
                     params=P(metric="affine2p", x=8, o1=4, e1=2, o2=24, e2=1, span="endsfree", tef=len(t8), heuristic="none"),
                     pattern=p8, text=t8, status=0, span=[78, 250, 0, 172]))
    # W9-W12 ends-free  wfaligner.rs:1263-1334
    kats.append(dict(id="W9_endsfree_global", src="src/wfaligner.rs:1263-1279",
                     params=P(x=6, o1=4, e1=2, span="endsfree", tef=len(t7)), pattern=p7, text=t7, status=0, score=-36,
                     cigar="13I9M1X12M15I"))
    p10, t10 = "AATTTAAGTCTGCTACTTTCACGCAGCT", "AATTTCAGTCTGGCTACTTTCACGTACGATGACAGACTCT"
    kats.append(dict(id="W10_endsfree_right_extent", src="src/wfaligner.rs:1281-1298",
                     params=P(x=6, o1=4, e1=2, span="endsfree", pef=len(p10), tef=len(t10)), pattern=p10, text=t10, status=0,
                     score=-24, cigar="5M1X6M1I11M4D1M15I"))
    p11 = "CTTTCACGTACGTGACAGTCTCT"
    kats.append(dict(id="W11_endsfree_left_extent", src="src/wfaligner.rs:1300-1316", params=P(x=6, o1=4, e1=2, span="endsfree"),
                     pattern=p11, text=t10, status=0, score=-48, cigar="16I12M1I6M1X4M"))
    p12, t12 = "CGCGTCTGACTGACTGACTAAACTTTCATGTACCTGACA", "AAACTTTCACGTACGTGACATATAGCGATCGATGACT"
    kats.append(dict(id="W12_endsfree_right_overlap", src="src/wfaligner.rs:1318-1334", params=P(x=6, o1=4, e1=2, span="endsfree"),
                     pattern=p12, text=t12, status=0, score=-92, cigar="19D9M1X4M1X5M17I"))
    # W13 clipping score  wfaligner.rs:1336-1381
    text_lf = "AAGGAGCTGAGAATTGTTCTTCCAGATACCTTTCCGACCTCTTCTTGGTT"
    text_rf = "GGAGTGCAGTGGTGCAATCTTGGCTCACTACAACCTCCGCATCCTGGGTT"
    pat_lf = "AAGGAGCTGAGAATTGTTCGTCCAGATACCTTTCCGACCTCTTCTTGGTT"
    pat_rf = "GGAGTGCAGTGGTGCAATCTTGGCTCACTACAACCTCTGCATCCTGGGTT"
    t13 = text_lf + "ATTT" * 10 + text_rf
    p13 = pat_lf + "ATTT" * 8 + pat_rf
    kats.append(dict(id="W13a_clipping_affine2p", src="src/wfaligner.rs:1336-1369",
                     params=P(metric="affine2p", x=8, o1=4, e1=2, o2=24, e2=1), pattern=p13, text=t13, status=0, score=-36,
                     cigar="19M1X62M8I37M1X12M", cigar_score=-36, clipped=[[50, -20]], clipped_cigar=[[50, "32M8I"]]))
    kats.append(dict(id="W13b_clipping_indel", src="src/wfaligner.rs:1371-1380", params=P(metric="indel", heuristic="none"),
                     pattern=p13, text=t13, status=0, score=12, cigar_score=12, clipped=[[19, 10], [0, 12]]))
    # W14 memory modes  wfaligner.rs:1383-1421
    for mem in ("high", "med", "low"):
        kats.append(dict(id="W14_memory_%s" % mem, src="src/wfaligner.rs:1383-1421",
                         params=P(metric="affine2p", x=8, o1=4, e1=2, o2=24, e2=1, memory=mem), pattern=PATTERN, text=TEXT,
                         status=0, score=-48, cigar="1M1X3M1I5M2X8M3I1M1X9M", cigar_score=-48, clipped=[[0, -48]]))
    # W15 BiWFA + heuristic  wfaligner.rs:1437-1454
    read = literal_after(wf, "fn test_invalid_sequence", "read")
    allele = literal_after(wf, "fn test_invalid_sequence", "allele")
    kats.append(dict(id="W15a_biwfa_default_heuristic", src="src/wfaligner.rs:1442-1448",
                     params=P(metric="affine2p", x=8, o1=4, e1=2, o2=24, e2=1, memory="ultralow"), pattern=read, text=allele,
                     status=-300, score=-2147483648))
    kats.append(dict(id="W15b_biwfa_no_heuristic", src="src/wfaligner.rs:1450-1453",
                     params=P(metric="affine2p", x=8, o1=4, e1=2, o2=24, e2=1, memory="ultralow", heuristic="none"), pattern=read,
                     text=allele, status=0, score=-881))
    # W16 SAM cigar  wfaligner.rs:1589-1676
    kats.append(dict(id="W16a_sam_identical", src="src/wfaligner.rs:1589-1634", params=P(x=4, o1=6, e1=2), pattern="TCTTTACTCTT",
                     text="TCTTTACTCTT", status=0, sam_true=[183], sam_false=[176]))
    kats.append(dict(id="W16b_sam_diff", src="src/wfaligner.rs:1636-1676", params=P(x=4, o1=6, e1=2, memory="low"),
                     pattern="TCTTTACTCTT", text="TCTTTACTATT", status=0, sam_true=[135, 24, 39], sam_false=[176]))
    # W17 get_alignment global  wfaligner.rs:1718-1752
    ops17 = "MXMMMIMMMMMXXMMMMMMMMIIIMXMMMMMMMMM"
    kats.append(dict(id="W17_get_alignment_global", src="src/wfaligner.rs:1718-1752", params=P(x=1, o1=5, e1=1), pattern=PATTERN,
                     text=TEXT, status=0, score=-18, ops=ops17, span=[0, 31, 0, 35]))
    # W18 get_alignment ends-free  wfaligner.rs:1794-1828
    t18 = "GGGGGGGGGGAGTGTCAATGGCTACGGGGGGGGGG"
    kats.append(dict(id="W18_get_alignment_ends_free", src="src/wfaligner.rs:1794-1828",
                     params=P(x=1, o1=5, e1=1, span="endsfree", tbf=len(t18), tef=len(t18)), pattern="AGTGTCAATGGCTAC", text=t18,
                     status=0, score=0, cigar="10I15M10I", span=[0, 15, 10, 25]))
    # W20 commented-out BiWFA test (informative): ops as W17, score() left at i32::MIN, cigar_score -18
    kats.append(dict(id="W20_biwfa_global_informative", src="src/wfaligner.rs:1754-1792 (commented out)",
                     params=P(x=1, o1=5, e1=1, memory="ultralow"), pattern=PATTERN, text=TEXT, status=0, score=-2147483648,
                     ops=ops17, cigar_score=-18, informative=True))
    json.dump(dict(source="PacificBiosciences/trgt v3.0.0 src/wfaligner.rs inline tests", kats=kats),
              open(os.path.join(HERE, "wfa_kats.json"), "w"), indent=1)

    # ------------------------------------------------------------------ HMM
    b = rs("src/hmm/builder.rs")
    q4 = re.search(r'fn parse_aga_repeat.*?let query = "([A-Z]+)"', b, re.S).group(1)
    hk = []
    hk.append(dict(id="H1_two_perfect_runs", src="src/hmm/builder.rs:208-216", motifs=["CAG", "A"], query="CAGCAGCAGCAGAAAAA",
                   remove_imperfect=False, summary=[[0, 12, 0], [12, 17, 1]]))
    hk.append(dict(id="H2_runs_separated_by_insertion", src="src/hmm/builder.rs:218-240", motifs=["CAG", "A"],
                   query="CAGCAGATCGATCGATCGATCGAAAAA", remove_imperfect=True,
                   summary=[[0, 6, 0], [6, 7, 1], [7, 10, 2], [10, 11, 1], [11, 14, 2], [14, 15, 1], [15, 18, 2], [18, 19, 1],
                            [19, 22, 2], [22, 27, 1]]))
    hk.append(dict(id="H3_imperfect_run", src="src/hmm/builder.rs:242-250", motifs=["CAG", "A"], query="CAGCAGCTGCAGCAGAAACAG",
                   remove_imperfect=False, summary=[[0, 15, 0], [15, 18, 1], [18, 21, 0]]))
    hk.append(dict(id="H4_aga_repeat", src="src/hmm/builder.rs:252-273", motifs=["AAG", "CAAC"], query=q4, remove_imperfect=True,
                   summary=[[0, 6, 2], [6, 14, 1], [14, 36, 2], [36, 93, 0], [93, 108, 2], [108, 111, 0], [111, 122, 2],
                            [122, 125, 0]]))
    hk.append(dict(id="H5_purity_perfect", src="src/hmm/purity.rs:48-55", motifs=["CAG", "CCG"], query="CAGCAGCAGCCGCCGCCGCCG",
                   purity=[1, 1]))
    hk.append(dict(id="H6_purity_imperfect", src="src/hmm/purity.rs:57-65", motifs=["CAG", "CCG"], query="CAGCGCAGCCGCCGCCGGG",
                   purity=[17, 20]))
    hk.append(dict(id="H7_purity_skip", src="src/hmm/purity.rs:67-75", motifs=["CAG", "CCG"], query="CAGCAGCAGTTTTTTTTCCGCCGCCG",
                   purity=[18, 26]))
    hk.append(dict(id="H8_purity_polyalanine", src="src/hmm/purity.rs:77-86", motifs=["GCN"], query="GCAGCCGCTGAG", purity=[11, 12]))
    hk.append(dict(id="H9_purity_empty", src="src/hmm/purity.rs:88-96", motifs=["CAG", "CCG"], query="", purity=None))
    hk.append(dict(id="H10_base_match_A", src="src/hmm/events.rs:124-129", motifs=["A"], state=3, base_match="A"))
    hk.append(dict(id="H11_base_match_N", src="src/hmm/events.rs:131-136", motifs=["N"], state=3, base_match="N"))
    # libm ln constants of SURVEY.md Appendix B.1a (glibc 2.35): detects libm drift
    ln = {"0.10": "-0x1.26bb1bbb55515p+1", "0.50": "-0x1.62e42fefa39efp-1", "0.90": "-0x1.af8e8210a415cp-4",
          "0.03": "-0x1.c0d6e3a1428a6p+1", "0.25": "-0x1.62e42fefa39efp+0", "0.75": "-0x1.269621134db92p-2",
          "(1.00-0.90)/2.00": "-0x1.7f7427b73e392p+1"}
    json.dump(dict(source="PacificBiosciences/trgt v3.0.0 src/hmm inline tests", kats=hk, ln_constants=ln),
              open(os.path.join(HERE, "hmm_kats.json"), "w"), indent=1)
    # E1: docs/tutorial.md:45 golden VCF sample column for example/
    e1 = dict(src="docs/tutorial.md:29-46", AL="33,33", ALLR="30-39,33-33", SD="15,14", MC="11,11", MS="0(0-33),0(0-33)",
              AP="1.000000,1.000000", ref_tr="CAG" * 20, alt="CAG" * 11)
    json.dump(e1, open(os.path.join(HERE, "example_e1.json"), "w"), indent=1)
    print("wrote", len(kats), "WFA KATs,", len(hk), "HMM KATs")


if __name__ == "__main__":
    main()
