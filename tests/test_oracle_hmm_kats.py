"""Pin the CPU oracle's HMM against every live known-answer test of the reference
(src/hmm/builder.rs:208-273, purity.rs:48-96, events.rs:124-136)."""
import json
import math
import os

import pytest

D = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hmm_kats.json")))
KATS = D["kats"]


def summarize(spans):
    out = []
    for m, s, e in [(int(a), int(b), int(c)) for a, b, c in spans]:
        if out and out[-1][2] == m:
            out[-1][1] = e
        else:
            out.append([s, e, m])
    return out


@pytest.mark.parametrize("kat", KATS, ids=[k["id"] for k in KATS])
def test_hmm_kat(oracle, kat):
    motifs = kat["motifs"]
    if "summary" in kat:
        path = oracle.hmm_label(motifs, kat["query"])
        if kat["remove_imperfect"]:
            path = oracle.hmm_remove_imperfect(motifs, path, kat["query"], 6)
        assert summarize(oracle.hmm_label_motifs(motifs, path)) == kat["summary"]
    elif "purity" in kat:
        q = kat["query"]
        path = oracle.hmm_label(motifs, q) if q else []
        pur, e, m = oracle.hmm_purity(motifs, path, q)
        if kat["purity"] is None:
            assert math.isnan(pur)
        else:
            num, den = kat["purity"]
            assert pur == num / den
    else:
        assert oracle.hmm_base_match(motifs, kat["state"]) == kat["base_match"]


def test_ln_constants_match_survey_table():
    # the host libm must reproduce these bit patterns
    for expr, want in D["ln_constants"].items():
        assert math.log(eval(expr)).hex() == float.fromhex(want).hex(), expr


def test_annotate_matches_pipeline_pieces(oracle):
    # label_with_hmm (tr.rs:454-492) == the composition of the individually pinned pieces
    motifs, q = ["CAG", "CCG"], "CAGCAGCAGTTTTTTTTCCGCCGCCG"
    a = oracle.hmm_annotate(motifs, q)
    path = oracle.hmm_label(motifs, q)
    assert list(a["path"]) == list(path)
    assert a["purity"] == 18.0 / 26.0
    sp = oracle.hmm_label_motifs(motifs, oracle.hmm_remove_imperfect(motifs, path, q, 6))
    kept = [s for s in sp.tolist() if s[0] < len(motifs)]
    assert list(a["counts"]) == [sum(1 for s in kept if s[0] == i) for i in range(len(motifs))]
    assert a["spans"].tolist() == [[0, 0, 9], [1, 17, 26]]


def test_annotate_empty_allele(oracle):
    a = oracle.hmm_annotate(["CAG"], "")
    assert len(a["path"]) == 0 and len(a["spans"]) == 0 and math.isnan(a["purity"]) and list(a["counts"]) == [0]
