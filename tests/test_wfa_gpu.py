"""Parity of the HIP wavefront aligner (trgt_wfa_batch through the C ABI) against the reference's own
known-answer tests and, on seeded random inputs, against the CPU oracle.  Bit-exact: status, score, CIGAR,
operation strings, count_matches and alignment spans."""
import json
import os

import numpy as np
import pytest

from helpers import mutate, rand_dna, repeat_allele

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wfa_kats.json")))["kats"]


@pytest.fixture(scope="module")
def W():
    from trgt_amd import wfaligner
    return wfaligner


def build(W, d):
    b = W.WFAligner.builder(W.AlignmentScope.Alignment if d["scope"] == "alignment" else W.AlignmentScope.Score,
                            {"high": W.MemoryModel.MemoryHigh, "med": W.MemoryModel.MemoryMed, "low": W.MemoryModel.MemoryLow,
                             "ultralow": W.MemoryModel.MemoryUltraLow}[d["memory"]])
    m = d["metric"]
    if m == "indel":
        b = b.indel()
    elif m == "edit":
        b = b.edit()
    elif m == "linear":
        b = b.linear(d["x"], d["e1"])
    elif m == "affine":
        b = b.affine(d["x"], d["o1"], d["e1"])
    else:
        b = b.affine2p(d["x"], d["o1"], d["e1"], d["o2"], d["e2"])
    if d["heuristic"] == "none":
        b = b.with_heuristic(W.Heuristic.none())
    return b.build()


@pytest.mark.parametrize("kat", KATS, ids=[k["id"] for k in KATS])
def test_reference_kat(W, kat):
    d = kat["params"]
    al = build(W, d)
    p, t = kat["pattern"].encode(), kat["text"].encode()
    if d["span"] == "endsfree":
        st = al.align_ends_free(p, d["pbf"], d["pef"], t, d["tbf"], d["tef"])
    else:
        st = al.align_end_to_end(p, t)
    assert int(st) == kat["status"]
    if "score" in kat:
        assert al.score() == kat["score"]
    if "cigar" in kat:
        assert al.cigar_string() == kat["cigar"]
    if "ops" in kat:
        assert al.cigar_operations().decode() == kat["ops"]
    if "span" in kat:
        (xs, xe), (ys, ye) = al.get_alignment_span()
        assert [xs, xe, ys, ye] == kat["span"]
    if "cigar_score" in kat:
        assert al.cigar_score() == kat["cigar_score"]
    for flank, want in kat.get("clipped", []):
        assert al.cigar_score_clipped(flank) == want
    for flank, want in kat.get("clipped_cigar", []):
        assert al.cigar_string(flank) == want
    if "sam_true" in kat:
        assert al.get_sam_cigar(True) == kat["sam_true"]
        assert al.get_sam_cigar(False) == kat["sam_false"]
        assert W.WFAligner.decode_sam_cigar(al.get_sam_cigar(True))[0][1] in "=X"


def _check_batch(oracle, got, op, pats, txts):
    n = len(pats)
    blob = b"".join(pats) + b"".join(txts)
    plen = np.array([len(x) for x in pats], np.uint32)
    tlen = np.array([len(x) for x in txts], np.uint32)
    pat_off = np.zeros(n, np.uint64)
    pat_off[1:] = np.cumsum(plen[:-1], dtype=np.uint64)
    txt_off = np.zeros(n, np.uint64)
    txt_off[1:] = np.cumsum(tlen[:-1], dtype=np.uint64)
    txt_off += np.uint64(int(plen.sum()))
    coff = got["cigar_off"]
    batch = dict(seqs=np.frombuffer(blob, np.uint8).copy(), pat_off=pat_off, pat_len=plen, txt_off=txt_off, txt_len=tlen,
                 cigar_off=coff, ops_off=coff)
    ref = oracle.wfa_batch(op, batch, n_threads=8)
    assert np.array_equal(got["status"], ref["status"])
    assert np.array_equal(got["score"], ref["score"])
    assert np.array_equal(got["n_match"], ref["n_match"])
    assert np.array_equal(got["span4"], ref["span4"])
    assert np.array_equal(got["cigar_len"], ref["cigar_len"])
    assert np.array_equal(got["ops_len"], ref["ops_len"])
    for j in range(n):
        o, cl, ol = int(coff[j]), int(ref["cigar_len"][j]), int(ref["ops_len"][j])
        assert np.array_equal(got["cigar"][o:o + cl], ref["cigar"][o:o + cl]), j
        assert bytes(got["ops"][o:o + ol]) == bytes(ref["ops"][o:o + ol]), j
    return ref


def test_flank_style_ends_free_random(oracle, W):
    # THREAD_WFA_FLANK: affine(2,5,1), Heuristic::None, MemoryHigh; align_ends_free(piece,0,0,read,|read|,|read|)
    rng = np.random.default_rng(20250509)
    pats, txts = [], []
    for i in range(160):
        flank = rand_dna(rng, 250)
        left, right = rand_dna(rng, int(rng.integers(0, 400))), rand_dna(rng, int(rng.integers(0, 600)))
        f = mutate(rng, flank, *([0.004, 0.002, 0.002] if i % 4 else [0.05, 0.02, 0.02]))
        read = left + f + right
        if i % 10 == 0:   # truncated read: only part of the flank is present
            cut = int(rng.integers(1, 250))
            read = f[cut:] + right if i % 20 == 0 else left + f[:cut]
        if i % 33 == 0:
            read = rand_dna(rng, int(rng.integers(1, 300)))  # flank absent
        pats.append(flank)
        txts.append(read)
    al = W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryHigh).affine(2, 5, 1).with_heuristic(W.Heuristic.none()).build()
    got = al.align_ends_free_batch(pats, 0, 0, txts, -1, -1)
    op = oracle.wfa_params(metric="affine", x=2, o1=5, e1=1, span="endsfree", pbf=0, pef=0, tbf=-1, tef=-1, heuristic="none")
    _check_batch(oracle, got, op, pats, txts)


@pytest.mark.parametrize("metric,pen", [("indel", {}), ("edit", {}), ("linear", dict(x=6, e1=2)), ("affine", dict(x=6, o1=4, e1=2)),
                                         ("affine", dict(x=2, o1=5, e1=1)), ("affine2p", dict(x=8, o1=4, e1=2, o2=24, e2=1))])
@pytest.mark.parametrize("heur", ["none", "default"])
def test_end_to_end_random_all_metrics(oracle, W, metric, pen, heur):
    rng = np.random.default_rng(hash((metric, heur)) % 2**32)
    pats, txts = [], []
    for i in range(60):
        a = rand_dna(rng, int(rng.integers(0, 300)))
        b = mutate(rng, a, 0.03, 0.02, 0.02) if i % 5 else rand_dna(rng, int(rng.integers(0, 120)))
        pats.append(a)
        txts.append(b)
    b_ = W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryHigh)
    b_ = {"indel": lambda: b_.indel(), "edit": lambda: b_.edit(), "linear": lambda: b_.linear(pen["x"], pen["e1"]),
          "affine": lambda: b_.affine(pen["x"], pen["o1"], pen["e1"]),
          "affine2p": lambda: b_.affine2p(pen["x"], pen["o1"], pen["e1"], pen["o2"], pen["e2"])}[metric]()
    if heur == "none":
        b_ = b_.with_heuristic(W.Heuristic.none())
    got = b_.build().align_end_to_end_batch(pats, txts)
    op = oracle.wfa_params(metric=metric, heuristic=heur, **pen)
    _check_batch(oracle, got, op, pats, txts)


@pytest.mark.parametrize("min_length", [100, 0])
def test_consensus_style_biwfa_random(oracle, W, min_length):
    # THREAD_WFA_CONSENSUS: BiWFA affine(2,5,1) + default wfadaptive heuristic, end-to-end, STR alleles
    rng = np.random.default_rng(99 + min_length)
    pats, txts = [], []
    for i in range(120):
        motif = [rand_dna(rng, int(rng.integers(2, 7)))]
        backbone = repeat_allele(rng, motif, int(rng.integers(20, 420)), err=0.0)
        read = mutate(rng, backbone, 0.01, 0.01, 0.01)
        if i % 3 == 0:  # stutter: +-1..3 motif copies
            k = len(motif[0]) * int(rng.integers(1, 4))
            read = read[:len(read) // 2] + (motif[0] * 3)[:k] + read[len(read) // 2:] if i % 2 else read[k:]
        pats.append(backbone)
        txts.append(read)
    al = W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryUltraLow).affine(2, 5, 1).build()
    p = al._params("end2end")
    p.bialign_min_length = min_length
    got = al._run_batch(p, pats, txts)
    op = oracle.wfa_params(metric="affine", x=2, o1=5, e1=1, memory="ultralow", heuristic="default", min_length=min_length)
    ref = _check_batch(oracle, got, op, pats, txts)
    assert (ref["status"] == 0).all()


def test_edit_distance_score_only_biwfa(oracle, W):
    # THREAD_WFA_ED: edit, Score scope, BiWFA + default heuristic (genotype_cluster.rs:238-248, len1*len2 <= 10000)
    rng = np.random.default_rng(4)
    pats, txts = [], []
    for i in range(200):
        a = rand_dna(rng, int(rng.integers(0, 100)))
        b = mutate(rng, a, 0.05, 0.03, 0.03) if i % 4 else rand_dna(rng, int(rng.integers(0, 100)))
        pats.append(a)
        txts.append(b)
    al = W.WFAligner.builder(W.AlignmentScope.Score, W.MemoryModel.MemoryUltraLow).edit().build()
    got = al.align_end_to_end_batch(pats, txts)
    op = oracle.wfa_params(metric="edit", scope="score", memory="ultralow", heuristic="default")
    ref = _check_batch(oracle, got, op, pats, txts)
    # sanity: plain DP edit distance on a few pairs
    for j in range(0, 200, 37):
        a, b = pats[j], txts[j]
        prev = list(range(len(b) + 1))
        for x in range(1, len(a) + 1):
            cur = [x] + [0] * len(b)
            for y in range(1, len(b) + 1):
                cur[y] = min(prev[y] + 1, cur[y - 1] + 1, prev[y - 1] + (a[x - 1] != b[y - 1]))
            prev = cur
        assert int(ref["score"][j]) == prev[len(b)]


def test_long_expanded_alleles_biwfa(oracle, W):
    rng = np.random.default_rng(8)
    pats, txts = [], []
    for n in (1500, 4000, 9000):
        bb = repeat_allele(rng, [b"GGCCTG"], n, err=0.0)
        pats.append(bb)
        txts.append(mutate(rng, bb, 0.004, 0.003, 0.003))
    al = W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryUltraLow).affine(2, 5, 1).build()
    got = al.align_end_to_end_batch(pats, txts)
    op = oracle.wfa_params(metric="affine", x=2, o1=5, e1=1, memory="ultralow", heuristic="default")
    _check_batch(oracle, got, op, pats, txts)


@pytest.mark.parametrize("pen", [(1, 0, 1), (1, 3, 1), (3, 2, 2), (4, 6, 2), (2, 5, 1), (5, 1, 3), (7, 9, 1)])
@pytest.mark.parametrize("span", ["end2end", "endsfree", "textfree"])
def test_dedicated_affine_kernel_penalties_and_shapes(oracle, W, pen, span):
    # exact unidirectional gap-affine batches run on wfa_fast_kernel: ring depth, "current level" sources (x = 1, o + e = 1, e = 1),
    # score-level parity (e > 1) and every ends-free shape are exercised here, next to degenerate sequences
    x, o, e = pen
    rng = np.random.default_rng(1000 * x + 100 * o + 10 * e + len(span))
    pats, txts = [], []
    for i in range(70):
        a = rand_dna(rng, int(rng.integers(1, 260)))
        if i % 7 == 0:
            b = rand_dna(rng, int(rng.integers(1, 200)))
        else:
            b = rand_dna(rng, int(rng.integers(0, 150))) + mutate(rng, a, 0.03, 0.015, 0.015) + rand_dna(rng, int(rng.integers(0, 150)))
        if i % 11 == 0:
            a, b = b, a            # pattern longer than text
        if i == 3:
            a, b = b"A", b"A"
        if i == 4:
            a, b = b"ACGTACGTAC", b"T"
        if i == 5:
            a, b = b"G" * 130, b"G" * 129 + b"C" + b"G" * 40   # long runs: many 4-byte windows per extension
        if span == "endsfree" and (len(a) < 6 or len(b) < 41):   # fixed free-end lengths must not exceed the sequences
            a, b = a + rand_dna(rng, 6), b + rand_dna(rng, 41)
        pats.append(bytes(a))
        txts.append(bytes(b))
    al = W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryHigh).affine(x, o, e).with_heuristic(W.Heuristic.none()).build()
    if span == "end2end":
        got = al.align_end_to_end_batch(pats, txts)
        op = oracle.wfa_params(metric="affine", x=x, o1=o, e1=e, heuristic="none")
    elif span == "textfree":
        got = al.align_ends_free_batch(pats, 0, 0, txts, -1, -1)
        op = oracle.wfa_params(metric="affine", x=x, o1=o, e1=e, span="endsfree", pbf=0, pef=0, tbf=-1, tef=-1, heuristic="none")
    else:
        got = al.align_ends_free_batch(pats, 3, 5, txts, 40, 7)
        op = oracle.wfa_params(metric="affine", x=x, o1=o, e1=e, span="endsfree", pbf=3, pef=5, tbf=40, tef=7, heuristic="none")
    _check_batch(oracle, got, op, pats, txts)


def _plain_affine_global(a, b, x, o, e):
    """Gotoh: the exact global gap-affine penalty, independent of any wavefront code."""
    INF = 1 << 28
    n, m = len(a), len(b)
    M = [0] + [INF] * m
    I = [INF] * (m + 1)   # gap in a (text base consumed)
    D = [INF] * (m + 1)
    for j in range(1, m + 1):
        I[j] = o + e * j
        M[j] = I[j]
    for i in range(1, n + 1):
        pM, pI, pD = M, I, D
        M, I, D = [INF] * (m + 1), [INF] * (m + 1), [INF] * (m + 1)
        D[0] = o + e * i
        M[0] = D[0]
        for j in range(1, m + 1):
            D[j] = min(pM[j] + o + e, pD[j] + e)
            I[j] = min(M[j - 1] + o + e, I[j - 1] + e)
            M[j] = min(pM[j - 1] + (0 if a[i - 1] == b[j - 1] else x), I[j], D[j])
    return M[m]


def test_biwfa_cigars_are_valid_and_optimal_without_the_oracle(W):
    """BiWFA tie-breaks are unpinned by the reference, so this test leaves the oracle out: whatever CIGAR the GPU returns
    for a completed end-to-end BiWFA alignment must (i) spell the text out of the pattern, (ii) cost what the aligner reports, and
    (iii) with Heuristic::None cost exactly the global gap-affine optimum computed by plain dynamic programming."""
    rng = np.random.default_rng(2718)
    pats, txts = [], []
    for i in range(90):
        motif = [rand_dna(rng, int(rng.integers(2, 7)))]
        backbone = repeat_allele(rng, motif, int(rng.integers(10, 330)), err=0.0)
        read = mutate(rng, backbone, 0.02, 0.015, 0.015)
        if i % 4 == 0:
            k = len(motif[0]) * int(rng.integers(1, 5))
            read = read[:len(read) // 3] + (motif[0] * 5)[:k] + read[len(read) // 3:] if i % 8 else read[k:]
        pats.append(backbone)
        txts.append(read)
    x, o, e = 2, 5, 1
    for heuristic in (W.Heuristic.none(), None):
        b = W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryUltraLow).affine(x, o, e)
        al = (b.with_heuristic(heuristic) if heuristic is not None else b).build()
        got = al.align_end_to_end_batch(pats, txts)
        for j, (p, t) in enumerate(zip(pats, txts)):
            if int(got["status"][j]) != 0:
                assert heuristic is None, "exact BiWFA must complete"
                continue
            ops = bytes(got["ops"][int(got["cigar_off"][j]):int(got["cigar_off"][j]) + int(got["ops_len"][j])]).decode()
            pi = ti = 0
            cost, prev = 0, ""
            for c in ops:
                if c in "MX":
                    assert (p[pi] == t[ti]) == (c == "M"), (j, pi, ti)
                    cost += x if c == "X" else 0
                    pi += 1; ti += 1
                elif c == "I":
                    cost += e + (o if prev != "I" else 0); ti += 1
                else:
                    assert c == "D"
                    cost += e + (o if prev != "D" else 0); pi += 1
                prev = c
            assert pi == len(p) and ti == len(t), (j, "the CIGAR does not span both sequences")
            if max(len(p), len(t)) > 100:  # (shorter pairs take the unidirectional base case, which never sets cigar.score)
                assert int(got["score"][j]) == -cost, (j, int(got["score"][j]), cost)
            if heuristic is not None:
                assert cost == _plain_affine_global(p, t, x, o, e), (j, cost)


@pytest.mark.parametrize("kb", [3, 7, 18])
def test_lds_arena_variant_of_the_biwfa_kernel_gives_the_same_results(W, kb):
    """TRGT_WFA_LDS=1: BiWFA batches of one wave per alignment first meet the LDS-arena variant of the kernel (wavefronts, descriptor
    rings and run-length buffers in LDS); what does not fit its budget -- here made small enough that much does not -- is redone by the
    HBM-arena variant.  Same engine code on the same inputs: status, score and CIGAR must not depend on where an alignment ran."""
    from trgt_amd import _lib
    rng = np.random.default_rng(77 + kb)
    pats, txts = [], []
    for i in range(400):
        n = int(rng.integers(1, 420)) if i % 9 else int(rng.integers(500, 1100))
        a = rand_dna(rng, n) if i % 5 else repeat_allele(rng, [b"CAG", b"CCG"], n, err=0.0)
        b = mutate(rng, a, *[(0.002, 0.001, 0.001), (0.02, 0.01, 0.01), (0.08, 0.04, 0.04)][i % 3])
        if i % 17 == 0:
            b = rand_dna(rng, int(rng.integers(1, 300)))  # unrelated: high scores, wide wavefronts
        pats.append(bytes(a)); txts.append(bytes(b) or b"A")
    lds = _lib.context_with_env(TRGT_WFA_LDS=1, TRGT_WFA_LDS_KB=kb, TRGT_WFA_LDS_SEQ=640)
    try:
        for build in (lambda c: W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryUltraLow).affine(2, 5, 1).build(ctx=c),
                      lambda c: W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryUltraLow).affine(2, 5, 1).with_heuristic(W.Heuristic.none()).build(ctx=c),
                      lambda c: W.WFAligner.builder(W.AlignmentScope.Score, W.MemoryModel.MemoryUltraLow).edit().build(ctx=c)):
            want = build(None).align_end_to_end_batch(pats, txts, want_ops=False)
            got = build(lds).align_end_to_end_batch(pats, txts, want_ops=False)
            assert np.array_equal(got["status"], want["status"]) and np.array_equal(got["score"], want["score"])
            assert np.array_equal(got["cigar_len"], want["cigar_len"]) and np.array_equal(got["n_match"], want["n_match"])
            for j in range(len(pats)):
                o, n = int(got["cigar_off"][j]), int(got["cigar_len"][j])
                assert np.array_equal(got["cigar"][o:o + n], want["cigar"][int(want["cigar_off"][j]):int(want["cigar_off"][j]) + n]), j
    finally:
        lds.close()
