"""The reference's own unit tests for the per-read steps either side of the hot path, transcribed as data by
tests/golden/make_read_kats.py and run through the NATIVE code the ingestion and the writers use (trgt_read_* / trgt_cigar_* /
trgt_median_i32 of include/trgt_hip.h are thin wrappers over those functions; host code, no GPU needed):
  clip_region.rs tests (7 calls), clip_bases.rs tests (15 calls), snp.rs extract_snps_offset on the example read's CIGAR, read.rs get_meth
  (MM / ML parsing), cigar.rs length helpers, utils/math.rs median (what simple_consensus uses), utils/region.rs (catalog coordinates)."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "read_kats.json")))


@pytest.fixture(scope="module")
def ro():
    from trgt_amd import readops
    return readops


def _same(got, exp):
    if exp is None:
        return got is None
    return got is not None and got["bases"].decode() == exp["bases"] and got["meth"] == exp["meth"] and got["cigar"] == exp["cigar"] and got["ref_pos"] == exp["ref_pos"] \
        and got["quals"] == b"(" * len(exp["bases"])


@pytest.mark.parametrize("case", KATS["clip_to_region"], ids=lambda c: "%s-%d-%d" % (c["test"], c["region_start"], c["region_end"]))
def test_clip_to_region(ro, case):
    r = case["read"]
    got = ro.clip_to_region(r["bases"].encode(), b"(" * len(r["bases"]), r["meth"], r["cigar"], r["ref_pos"], (case["region_start"], case["region_end"]))
    assert _same(got, case["expected"]), (got, case)


@pytest.mark.parametrize("case", KATS["clip_bases"], ids=lambda c: "%s-%d-%d" % (c["test"], c["left_len"], c["right_len"]))
def test_clip_bases(ro, case):
    r = case["read"]
    got = ro.clip_bases(r["bases"].encode(), b"(" * len(r["bases"]), r["meth"], r["cigar"], r["ref_pos"], case["left_len"], case["right_len"])
    assert _same(got, case["expected"]), (got, case)


def test_extract_snps_offset(ro):
    for c in KATS["extract_snps_offset"]:
        assert ro.extract_snps_offset(c["cigar"], c["ref_pos"], c["region"][0], c["region"][1]) == c["expected"]
    # hand-made: X runs inside [start, end] are not reported, the ones beyond the end count from the end
    assert ro.extract_snps_offset("1M1X1M1X1M1X1M1X1M", 0, 2, 6) == [-1, 1]


@pytest.mark.parametrize("case", KATS["get_meth"], ids=lambda c: c["test"])
def test_get_meth(ro, case):
    assert ro.get_meth(case["bases"].encode(), case["mm"], case["ml"], case["reverse"]) == case["expected"]


def test_get_meth_numeric_code_and_multi_code_entries(ro):
    # an all-digit ChEBI code is ONE modification: the ML values of a later "C+m" entry must not be misaligned by it
    bases = b"ACGTCGACG"  # C at 1, 4, 7: all CpGs
    assert ro.get_meth(bases, "C+76792,0;C+m,0,1;", [9, 200, 100]) == [200, 0, 100]
    assert ro.get_meth(bases, "C+hm,0,1;", [1, 50, 2, 60]) == [50, 0, 60]  # interleaved per code: m is the second code
    assert ro.get_meth(bases, "A+a,0;", [7]) is None                          # no C call at all -> None


def test_cigar_length_helpers(ro):
    for c in KATS["cigar_lens"]:
        if "ops" in c:
            assert ro.query_len([tuple(o) for o in c["ops"]]) == c["expected"]
        else:
            assert getattr(ro, c["fn"])(tuple(c["op"])) == c["expected"], c


@pytest.mark.parametrize("case", KATS["median"], ids=lambda c: c["test"])
def test_median(ro, case):
    assert ro.median(case["data"]) == case["expected"]


def test_catalog_coordinates_parse_like_genomic_region(tmp_path):
    """utils/region.rs tests through the catalog reader (the BED line's three fields are the "contig:start-end" encoding, locus.rs:49): a
    line that does not parse is skipped with the reference's message and the next line is read (locus.rs:93-137)."""
    from trgt_amd import ingest
    ex = os.path.join(HERE, "golden", "example")
    info = "ID=x;MOTIFS=CAG;STRUC=(CAG)n"
    lines, want = [], []
    for c in KATS["region"]:
        contig, rest = c["encoding"].rsplit(":", 1) if c["ok"] else c["encoding"].rsplit(":", 1)
        start, end = rest.split("-")
        lines.append("%s\t%s\t%s\t%s" % (contig, start, end, info))
        want.append(None if c["ok"] else c["error"])
    lines += ["", "chrA\t12abc\t20\t" + info, "chrA\t-3\t20\t" + info, "chrA\t10001\t10061\tID=x;MOTIFS=CAG", "chrA\t250\t260\t" + info, "chrA\t251\t260\t" + info]
    bed = tmp_path / "cat.bed"
    bed.write_text("\n".join(lines) + "\n")
    rd = ingest.Reader(os.path.join(ex, "sample.bam"), os.path.join(ex, "reference.fasta"))
    b = rd.batch(str(bed))
    msgs = b["skipped"]
    # region.rs vectors: the valid one names a contig the example genome lacks (skipped for THAT reason), the others carry their message
    assert msgs[0] == "Error at BED line 1: FASTA reference does not contain chromosome 'chr1' in BED file"
    for i, w in enumerate(want[1:], start=2):
        assert msgs[i - 1] == "Error at BED line %d: %s" % (i, w)
    n = len(KATS["region"])
    assert msgs[n].startswith("Error at BED line %d: Expected 4 fields in the format 'chrom start end info', found 0" % (n + 1))  # a blank line is a line
    assert msgs[n + 1] == "Error at BED line %d: Invalid region encoding: chrA:12abc-20" % (n + 2)   # u32::parse refuses trailing text ...
    assert msgs[n + 2] == "Error at BED line %d: Invalid region encoding: chrA:-3-20" % (n + 3)      # ... and a sign (four elements after the split)
    assert msgs[n + 3] == "Error at BED line %d: STRUC field missing" % (n + 4)
    assert msgs[n + 4] == "Error at BED line %d: Region start '250' with flank length '250' underflows for chromosome 'chrA'." % (n + 5)  # start < flank_len + 1
    assert len(msgs) == n + 5 and b["n_loci"] == 1 and int(b["region_start"][0]) == 251  # the last line is a locus
