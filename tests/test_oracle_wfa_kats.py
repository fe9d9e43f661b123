"""Pin the CPU oracle's WFA against every known-answer test the reference holds for this path
(PacificBiosciences/trgt src/wfaligner.rs:1136-1828)."""
import json
import os

import pytest

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wfa_kats.json")))["kats"]


def _params(oracle, d):
    heur = d["heuristic"]
    return oracle.wfa_params(metric=d["metric"], x=d["x"], o1=d["o1"], e1=d["e1"], o2=d["o2"], e2=d["e2"], span=d["span"],
                             pbf=d["pbf"], pef=d["pef"], tbf=d["tbf"], tef=d["tef"], scope=d["scope"], memory=d["memory"],
                             heuristic=heur)


@pytest.mark.parametrize("kat", KATS, ids=[k["id"] for k in KATS])
def test_wfa_kat(oracle, kat):
    p = _params(oracle, kat["params"])
    r = oracle.wfa_align(p, kat["pattern"].encode(), kat["text"].encode())
    assert r["status"] == kat["status"]
    if "score" in kat:
        assert r["score"] == kat["score"]
    if "cigar" in kat:
        assert oracle.cigar_string(r["ops"]) == kat["cigar"]
    if "ops" in kat:
        assert r["ops"] == kat["ops"]
    if "span" in kat:
        assert r["span"] == kat["span"]
    if "cigar_score" in kat:
        assert oracle.cigar_score(p, r["ops"]) == kat["cigar_score"]
    for flank, want in kat.get("clipped", []):
        assert oracle.cigar_score_clipped if False else True
        assert oracle.cigar_score(p, r["ops"], clipped=flank) == want
    for flank, want in kat.get("clipped_cigar", []):
        assert oracle.cigar_string(r["ops"][flank:len(r["ops"]) - flank]) == want
    if "sam_true" in kat:
        assert oracle.cigar_rle(r["ops"], True) == kat["sam_true"]
        assert oracle.cigar_rle(r["ops"], False) == kat["sam_false"]


def test_count_matches_and_span_semantics(oracle):
    # wfaligner.rs:864-908 / 988-1000 on W18: 10I15M10I
    kat = [k for k in KATS if k["id"].startswith("W18")][0]
    r = oracle.wfa_align(_params(oracle, kat["params"]), kat["pattern"].encode(), kat["text"].encode())
    assert r["n_match"] == 15
