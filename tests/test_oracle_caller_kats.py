"""The reference's own known-answer tests for the callers of the hot path and the wrapper's accessors (tests/golden/caller_kats.json,
made by tests/golden/make_caller_kats.py from haploid.rs:36-61, diploid.rs:109-120, span_locater.rs:72-130, events.rs:124-145,
wfaligner.rs:1423-1587) against the CPU oracle and the host-side mirror."""
import json
import os

import numpy as np
import pytest

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "caller_kats.json")))


@pytest.mark.parametrize("kat", KATS["genotype"], ids=lambda k: k["id"])
def test_length_genotyper_kats(oracle, kat):
    got = oracle.genotype_sizes(kat["ploidy"], kat["sizes"], kat["counts"])
    assert got == [(e["size"], tuple(e["ci"])) for e in kat["expected"]]


def test_exact_flank_search_kats(oracle):
    # windows().position(): the leftmost exact occurrence; a piece longer than the sequence or absent from it is None (the
    # wavefront fallback of find_spans then rejects these: nothing of the piece matches)
    for c in KATS["exact_search"]["cases"]:
        piece, seq = c["piece"].encode(), c["seq"].encode()
        blob = np.frombuffer(seq, np.uint8).copy() if seq else np.zeros(1, np.uint8)
        start, end, used = np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32)
        import ctypes as C
        cells = C.c_int64()
        rc = oracle.lib().orc_find_spans(oracle._p(np.frombuffer(piece, np.uint8).copy()), len(piece), C.c_int64(1), oracle._p(blob),
                                         oracle._p(np.zeros(1, np.uint64)), oracle._p(np.array([len(seq)], np.uint32)), 2, 5, 1,
                                         C.c_double(len(piece) * 0.7), oracle._p(start), oracle._p(end), oracle._p(used), C.byref(cells))
        assert rc == 0
        exp = c["span"]
        if exp is None:
            assert (int(start[0]), int(end[0])) == (-1, -1), c
        else:
            assert [int(start[0]), int(end[0])] == exp and int(used[0]) == 0, c


def test_base_match_kats(oracle):
    for c in KATS["base_match"]["cases"]:
        if "motifs" in c:
            blob, off = oracle.motif_blob(c["motifs"])
            got = chr(oracle.lib().orc_hmm_base_match(oracle._p(blob), oracle._p(off), len(c["motifs"]), c["state"]))
        else:
            got = oracle.hmm_base_match_ems(c["ems"], c["state"])
        assert got == c["expected"], c["id"]


def test_wrapper_accessor_kats():
    # WFAligner::builder(..).<penalties>().build().get_penalties(), with or without a heuristic; set_heuristic takes every variant
    from trgt_amd.wfaligner import AlignmentScope, Heuristic, MemoryModel, WFAligner
    names = {"match_": "match_", "mismatch": "mismatch", "indel": "indel", "gap_opening": "gap_opening", "gap_extension": "gap_extension",
             "gap_opening1": "gap_opening", "gap_extension1": "gap_extension", "gap_opening2": "gap_opening2", "gap_extension2": "gap_extension2"}
    ctor = {"WFadaptive": Heuristic.wfadaptive, "WFmash": Heuristic.wfmash, "XDrop": Heuristic.xdrop, "ZDrop": Heuristic.zdrop,
            "BandedStatic": Heuristic.banded_static, "BandedAdaptive": Heuristic.banded_adaptive, "None": Heuristic.none}
    for c in KATS["wrapper_accessors"]["penalties"]:
        b = WFAligner.builder(AlignmentScope.Alignment, MemoryModel.MemoryLow)
        b = getattr(b, c["builder"])(*c["args"])
        if c["heuristic"]:
            b = b.with_heuristic(ctor[c["heuristic"][0]](*c["heuristic"][1:]))
        a = b.build()
        pen = a.get_penalties()
        assert pen.kind == c["kind"], c
        for k, v in c["fields"].items():
            assert getattr(pen, names[k]) == v, (c, k)
        if c["heuristic"]:
            assert a.get_heuristics() == ctor[c["heuristic"][0]](*c["heuristic"][1:])
    a = WFAligner.builder(AlignmentScope.Alignment, MemoryModel.MemoryHigh).affine(6, 4, 2).build()
    assert a.get_heuristics() == Heuristic.wfadaptive(10, 50, 1)  # wavefront_aligner_attr_default
    for h in KATS["wrapper_accessors"]["set_heuristic"]:
        a.set_heuristic(ctor[h["kind"]](*h["args"]))
        assert a.get_heuristics().kind == h["kind"] and list(a.get_heuristics().args) == h["args"]


def decode_flank_read(enc):
    """the read encoding of genotype_flank.rs:297-337 (see tests/golden/make_caller_kats.py): (repeat bases, mismatch offsets, start, end)"""
    idx = [i for i, c in enumerate(enc) if c in "ACGT"]
    s, e = idx[0], idx[-1] + 1
    mm = [(i - s) if i < s else (i - e) for i, c in enumerate(enc) if c == "X"]
    return enc[s:e].encode(), mm, -s, len(enc) - e


def test_genotype_flank_kats(oracle):
    for kat in KATS["genotype_flank"]:
        rd = [decode_flank_read(e) for e in kat["reads"]]
        got = oracle.genotype_flank([r[0] for r in rd], None, [r[2] for r in rd], [r[3] for r in rd], [r[1] for r in rd])
        exp = kat["expected"]
        if exp is None:
            assert got is None, kat["id"]
        else:
            assert got == dict(gt=[(g["size"], tuple(g["ci"])) for g in exp["gt"]], alleles=exp["alleles"], assignment=exp["assignment"]), kat["id"]


def test_genotype_flank_with_haplotype_tags(oracle):
    # get_trs_with_hp (genotype_flank.rs:43-76): >= 70 % of the reads tagged, both haplotypes seen; untagged reads alternate
    trs = [b"CAGCAGCAG", b"CAGCAGCAG", b"CAGCAGCAGCAG", b"CAGCAGCAGCAG", b"CAGCAGCAGCAG", b"CAGCAG", b"CAGCAGCAGCAGCAG"]
    hp = [2, 2, 1, 1, 1, None, None]
    z = [0] * len(trs)
    got = oracle.genotype_flank(trs, hp, z, z, [[] for _ in trs])
    # haplotype 1 -> allele index 0 = {12, 12, 12} + untagged #1 (tie-breaker 0), haplotype 2 -> {9, 9} + untagged #2; smaller first
    assert got["alleles"] == ["CAGCAGCAG", "CAGCAGCAGCAG"] and got["assignment"] == [0, 0, 1, 1, 1, 1, 0]
    assert got["gt"] == [(9, (9, 15)), (12, (6, 12))]
    assert oracle.genotype_flank(trs, [1, 1, None, None, None, None, 2], z, z, [[] for _ in trs]) is None  # 3 of 7 tagged
    assert oracle.genotype_flank(trs[:3], [1, 1, 1], z[:3], z[:3], [[], [], []]) is None                    # one haplotype only
