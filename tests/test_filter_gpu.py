"""The register-resident pre-filter of the flank fallback alignments (trgt_amd/csrc/wfa_reg.hip) against the oracle's exact
WFA (oracle/wfa.cpp): the optimal score and the number of wavefront offsets must be WFA2-lib's, bit for bit; the match
bound must never be below count_matches() of the back-traced alignment (span_locater.rs:17-22 only compares it with a threshold)."""
import numpy as np
import pytest

from helpers import mutate, rand_dna

pytestmark = pytest.mark.gpu


def _oracle(oracle, pats, txts, scoring=(2, 5, 1)):
    out = []
    for p, t in zip(pats, txts):
        pp = oracle.wfa_params(metric="affine", x=scoring[0], o1=scoring[1], e1=scoring[2], span="endsfree", pbf=0, pef=0, tbf=len(t), tef=len(t), heuristic="none")
        out.append(oracle.wfa_align(pp, p, t))
    return out


def _check(oracle, pats, txts, min_matches=175, judged=True, scoring=(2, 5, 1), early_reject=False):
    from trgt_amd.wfaligner import flank_filter_batch
    r = flank_filter_batch(pats, txts, min_matches, scoring=scoring, early_reject=early_reject)
    ref = _oracle(oracle, pats, txts, scoring)
    cells = 0
    for j, o in enumerate(ref):
        assert o["status"] == 0
        if judged:
            assert int(r["score"][j]) == o["score"], (j, int(r["score"][j]), o["score"], len(pats[j]), len(txts[j]))
            assert int(r["bound"][j]) >= o["n_match"], (j, int(r["bound"][j]), o["n_match"])
            assert int(r["keep"][j]) == (1 if int(r["bound"][j]) >= min_matches else 0)
            cells += o["cells"]
        if o["n_match"] >= min_matches:
            assert int(r["keep"][j]) == 1, (j, o["n_match"])  # never reject what the reference accepts
    if judged:
        assert r["offsets"] == cells
    return r, ref


def _flank_jobs(rng, n, tlen_lo=300, tlen_hi=1000, plen=250):
    """Flank pieces against reads: complete copies with errors, copies cut at either end of the read, absent flanks."""
    pats, txts = [], []
    for _ in range(n):
        p = rand_dna(rng, plen)
        tl = int(rng.integers(tlen_lo, tlen_hi + 1))
        kind = int(rng.integers(0, 6))
        body = mutate(rng, p, *(float(x) for x in rng.choice([0.002, 0.01, 0.03, 0.08], 3)))
        if kind == 0:    # somewhere inside
            a = int(rng.integers(0, max(1, tl - len(body))))
            t = rand_dna(rng, a) + body + rand_dna(rng, max(0, tl - a - len(body)))
        elif kind == 1:  # cut by the end of the read: a prefix of the flank survives
            keep = int(rng.integers(1, plen))
            t = rand_dna(rng, max(plen, tl - keep)) + body[:keep]
        elif kind == 2:  # cut by the start of the read: a suffix survives
            keep = int(rng.integers(1, plen))
            t = body[len(body) - keep:] + rand_dna(rng, max(plen, tl - keep))
        elif kind == 3:  # absent
            t = rand_dna(rng, tl)
        elif kind == 4:  # low-complexity: ties everywhere
            unit = rand_dna(rng, int(rng.integers(1, 5)))
            p = (unit * plen)[:plen]
            t = mutate(rng, (unit * tl)[:tl], 0.02, 0.01, 0.01)
        else:            # flank with a long insertion / deletion inside
            cutp = int(rng.integers(40, plen - 40))
            if rng.random() < 0.5:
                body = body[:cutp] + rand_dna(rng, int(rng.integers(5, 120))) + body[cutp:]
            else:
                body = body[:cutp] + body[cutp + int(rng.integers(5, 100)):]
            a = int(rng.integers(0, max(1, tl - len(body))))
            t = rand_dna(rng, a) + body + rand_dna(rng, max(0, tl - a - len(body)))
        if len(t) < plen:
            t = t + rand_dna(rng, plen - len(t))
        pats.append(p); txts.append(t)
    return pats, txts


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_filter_matches_oracle_on_flank_like_jobs(oracle, seed):
    rng = np.random.default_rng(seed)
    pats, txts = _flank_jobs(rng, 160)
    r, ref = _check(oracle, pats, txts)
    assert 0 < int(r["keep"].sum()) < len(pats)


def test_filter_all_instantiations_and_short_patterns(oracle):
    rng = np.random.default_rng(7)
    for plen, lo, hi in ((250, 250, 760), (250, 780, 1020), (250, 1040, 1280), (254, 300, 900), (31, 31, 400), (1, 1, 64), (100, 100, 1400)):
        pats, txts = _flank_jobs(rng, 40, lo, hi, plen) if plen >= 100 else ([rand_dna(rng, plen) for _ in range(40)], [rand_dna(rng, int(rng.integers(lo, hi + 1))) for _ in range(40)])
        _check(oracle, pats, txts, min_matches=max(1, int(plen * 0.7)))


def test_filter_text_end_region(oracle):
    """Diagonals k > tlen - plen: offsets pass the end of the text, M cells are nulled and I cells trimmed there."""
    rng = np.random.default_rng(11)
    pats, txts = [], []
    for _ in range(120):
        p = rand_dna(rng, 250)
        tl = int(rng.integers(250, 330))
        keep = int(rng.integers(100, 250))
        t = rand_dna(rng, tl - keep) + mutate(rng, p, 0.02, 0.02, 0.02)[:keep]
        pats.append(p); txts.append(t if len(t) >= 250 else t + rand_dna(rng, 250 - len(t)))
    _check(oracle, pats, txts)


def test_filter_keeps_what_it_cannot_judge(oracle):
    from trgt_amd.wfaligner import flank_filter_batch
    rng = np.random.default_rng(5)
    p = rand_dna(rng, 250)
    jobs = [(p, rand_dna(rng, 100)),            # text shorter than the pattern
            (rand_dna(rng, 300), rand_dna(rng, 800)),   # pattern longer than 254
            (p, rand_dna(rng, 2000)),           # more diagonals than the largest instantiation
            (p, rand_dna(rng, 400) + b"\x01" + rand_dna(rng, 300)),  # a byte that collides with a sentinel
            (p, rand_dna(rng, 700))]            # an ordinary one
    r = flank_filter_batch([a for a, _ in jobs], [b for _, b in jobs], 175)
    assert list(r["keep"][:4]) == [1, 1, 1, 1] and list(r["bound"][:4]) == [-1] * 4
    assert int(r["score"][4]) == _oracle(oracle, [jobs[4][0]], [jobs[4][1]])[0]["score"] and int(r["keep"][4]) == 0


def test_filter_on_synthetic_batch_decisions(oracle):
    """The bench workload's own fallback alignments: every alignment the reference accepts is kept, and the kept share is small."""
    from trgt_amd import synth
    b = synth.generate(120, first_locus=1000)
    pats, txts = [], []
    for l in range(b["n_loci"]):
        a0, a1 = int(b["locus_read_begin"][l]), int(b["locus_read_begin"][l + 1])
        lf = bytes(b["flank_blob"][int(b["lf_off"][l]):int(b["lf_off"][l]) + int(b["lf_len"][l])])[-250:]
        rf = bytes(b["flank_blob"][int(b["rf_off"][l]):int(b["rf_off"][l]) + int(b["rf_len"][l])])[:250]
        for r in range(a0, a1):
            rd = bytes(b["read_blob"][int(b["read_off"][r]):int(b["read_off"][r]) + int(b["read_len"][r])])
            if not b["read_truncated"][r]:
                continue
            for piece in (lf, rf):
                if rd.find(piece) < 0 and len(rd) >= 250:
                    pats.append(piece); txts.append(rd)
    assert len(pats) > 100
    r, ref = _check(oracle, pats, txts)
    exact_keep = sum(1 for o in ref if o["n_match"] >= 175)
    assert int(r["keep"].sum()) <= exact_keep + max(3, len(pats) // 20)   # the bound is tight where it matters


@pytest.mark.parametrize("seed", [11, 12])
def test_early_rejection_never_drops_an_accepted_alignment(oracle, seed):
    # with early_reject the kernel stops an alignment once no wavefront cell can still reach min_matches: exact rejections (the
    # reference's count_matches is below the threshold for every one of them), exact scores for all the others, fewer offsets
    from trgt_amd.wfaligner import flank_filter_batch
    rng = np.random.default_rng(seed)
    pats, txts = _flank_jobs(rng, 200)
    ref = _oracle(oracle, pats, txts)
    full = flank_filter_batch(pats, txts, 175)
    for mm in (175, 100, 240, 0):
        r = flank_filter_batch(pats, txts, mm, early_reject=True)
        n_early = 0
        for j, o in enumerate(ref):
            if int(r["score"][j]) == -2**31 + 1:
                n_early += 1
                assert int(r["keep"][j]) == 0 and o["n_match"] < mm, (j, o["n_match"], mm)
            else:
                assert int(r["score"][j]) == o["score"] and int(r["bound"][j]) >= o["n_match"]
                assert int(r["keep"][j]) == (1 if int(r["bound"][j]) >= mm else 0)
            if o["n_match"] >= mm:
                assert int(r["keep"][j]) == 1
        assert r["offsets"] <= full["offsets"]
        if mm == 175:
            assert n_early > 0 and r["offsets"] < full["offsets"]
        if mm == 0:
            assert n_early == 0 and r["offsets"] == full["offsets"]


def test_two_launches_by_text_length_and_forced_instantiations(oracle, monkeypatch):
    # a batch large enough to run as two launches (texts above / up to 1024 diagonals) must judge every job exactly as one launch does;
    # so must every instantiation when it is forced onto the whole batch (TRGT_FILTER_FORCE: jobs it cannot hold are kept unseen)
    from trgt_amd.wfaligner import flank_filter_batch
    rng = np.random.default_rng(77)
    pats, txts = _flank_jobs(rng, 1400, tlen_lo=300, tlen_hi=890)
    base, ref = _check(oracle, pats, txts)  # (>= 1024 jobs, longest text above 773 bases: two launches)
    from trgt_amd import _lib
    one = flank_filter_batch(pats, txts, 175, ctx=_lib.context_with_env(TRGT_FILTER_ONE_LAUNCH=1))
    for k in ("score", "bound", "keep"):
        assert np.array_equal(base[k], one[k]), k
    assert base["offsets"] == one["offsets"]
    for force, max_diag in ((91, 1152), (52, 1280), (42, 1024), (71, 896)):
        r = flank_filter_batch(pats, txts, 175, ctx=_lib.context_with_env(TRGT_FILTER_FORCE=force))  # (developer switches: contexts created by the developer build)
        for j in range(len(pats)):
            if len(pats[j]) + len(txts[j]) + 1 <= max_diag:
                assert (int(r["score"][j]), int(r["bound"][j]), int(r["keep"][j])) == (int(base["score"][j]), int(base["bound"][j]), int(base["keep"][j])), (force, j)
            else:
                assert int(r["keep"][j]) == 1, (force, j)


@pytest.mark.parametrize("seed", [11, 12])
def test_targeted_preset_1_0_1_has_a_filter_too(oracle, seed):
    """--aln-scoring 1,0,1 (the targeted preset, cli.rs:271-280): X = OE = 1, one ring level updated in place.  Exact score, offset count
    and a bound that never undercuts count_matches(), on the same job mix and on every instantiation; and what the early rejection drops
    the reference would not have accepted."""
    rng = np.random.default_rng(seed)
    pats, txts = _flank_jobs(rng, 120)
    r, ref = _check(oracle, pats, txts, scoring=(1, 0, 1))
    assert 0 < int(r["keep"].sum()) < len(pats)
    for plen, lo, hi in ((250, 780, 1000), (250, 1040, 1150), (254, 1100, 1160), (200, 200, 700)):
        p2, t2 = _flank_jobs(rng, 24, lo, hi, plen)
        _check(oracle, p2, t2, min_matches=int(plen * 0.7), scoring=(1, 0, 1))
    from trgt_amd.wfaligner import flank_filter_batch
    e = flank_filter_batch(pats, txts, 175, scoring=(1, 0, 1), early_reject=True)
    for j, o in enumerate(ref):
        if o["n_match"] >= 175:
            assert int(e["keep"][j]) == 1, j
        if int(e["keep"][j]) and int(e["score"][j]) > -(1 << 30):
            assert int(e["score"][j]) == o["score"], j
    assert e["offsets"] <= r["offsets"]
    with pytest.raises(Exception):
        flank_filter_batch(pats[:2], txts[:2], 175, scoring=(3, 1, 1))
