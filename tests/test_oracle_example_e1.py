"""Pins the oracle's whole locus pipeline (find_tr_spans -> genotype -> consensus -> HMM labels) against the one
end-to-end golden the reference publishes: the VCF record of its example data set (docs/tutorial.md:29-46, SURVEY.md
Appendix E row E1).  Inputs: tests/golden/example_e1_reads.json = the reads of example/sample.bam clipped exactly as
analyze_tr clips them (made by tests/golden/make_example_fixture.py); expected: tests/golden/example_e1.json."""
import json
import os

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_e1():
    locus = json.load(open(os.path.join(GOLD, "example_e1_reads.json")))["loci"][0]
    want = json.load(open(os.path.join(GOLD, "example_e1.json")))
    return locus, want


def test_fixture_shape():
    locus, want = load_e1()
    assert locus["tr"] == want["ref_tr"] and locus["motifs"] == ["CAG"]
    assert len(locus["left_flank"]) == 250 and len(locus["right_flank"]) == 250
    assert len(locus["reads"]) == 33 and all(set(r) <= set("ACGT") for r in locus["reads"])


def test_oracle_reproduces_tutorial_vcf_record(oracle):
    locus, want = load_e1()
    r = oracle.locus_analyze(locus["left_flank"].encode(), locus["right_flank"].encode(), locus["tr"].encode(),
                             [m.encode() for m in locus["motifs"]], [x.encode() for x in locus["reads"]])
    assert r["alleles"] == [want["alt"], want["alt"]]
    for k in ("AL", "ALLR", "SD", "MC", "MS", "AP"):
        assert r[k] == want[k], k
