"""Reads beyond the dedicated kernels' texts (kilobases: expanded alleles): their flank alignments meet the pre-filter window by window
(trgt_amd/csrc/spans.hip, LongWinArgs) before the exact kernel, and what the windows keep is aligned again inside the band they name (heavy_band_kernel).  The shortcuts must never show: hand-made long reads -- flanks with errors
around the acceptance threshold, flanks near the borders of the filter's windows, flanks split by an insertion, flanks cut by the end of
the read, reads without a flank, two copies of a flank -- are compared read by read with the oracle's find_tr_spans, and with the same
library run without the window filter."""
import numpy as np
import pytest

from helpers import rand_dna

pytestmark = pytest.mark.gpu

WL, STEP = 1285, 1285 - 513  # window length and distance for 250-base flanks (spans.hip)


def _mutate(rng, seq, n_sub=0, n_del=0, n_ins=0):
    b = bytearray(seq)
    for p in sorted(rng.choice(len(b), size=n_sub, replace=False).tolist()) if n_sub else []:
        b[p] = int(rng.choice([c for c in b"ACGT" if c != b[p]]))
    for _ in range(n_del):
        p = int(rng.integers(1, len(b) - 1)); del b[p]
    for _ in range(n_ins):
        p = int(rng.integers(1, len(b) - 1)); b.insert(p, int(rng.choice(list(b"ACGT"))))
    return bytes(b)


def _spans_equal(oracle, loci, env_ctx):
    from trgt_amd import _lib, locus
    b = locus.pack(loci)
    got = locus.find_tr_spans_batch(b)
    plain = locus.find_tr_spans_batch(b, ctx=env_ctx)
    for x, y, name in zip(got, plain, ("span_start", "span_end", "lf_hit", "rf_hit")):
        assert np.array_equal(x, y), name
    # what the window filter keeps is back-traced inside the band its windows name (penalty and end diagonal: the smallest over the
    # windows); without the band, and with bands that take only some of the alignments, the same spans
    for band in ("0", "12", "256", "no seed search"):
        # ("no seed search": the long reads' alignments straight to the window filter, without the shortcuts and seeded windows in front)
        bctx = _lib.context_with_env(TRGT_HEAVY_BAND=band) if band != "no seed search" else _lib.context_with_env(TRGT_NO_LONG_WINDOW=1)
        try:
            other = locus.find_tr_spans_batch(b, ctx=bctx)
        finally:
            bctx.close()
        for x, y, name in zip(got, other, ("span_start", "span_end", "lf_hit", "rf_hit")):
            assert np.array_equal(x, y), (band, name)
    r = 0
    n_some = 0
    for L in loci:
        ref = oracle.locus_analyze(L["left_flank"], L["right_flank"], L["tr"], L["motifs"], L["reads"])  # (find_tr_spans is its first step)
        n = len(L["reads"])
        assert np.array_equal(got[0][r:r + n], ref["span_start"]) and np.array_equal(got[1][r:r + n], ref["span_end"]), (r, [len(x) for x in L["reads"]])
        n_some += int((np.asarray(ref["span_start"]) >= 0).sum())
        r += n
    return n_some


def test_long_reads_against_the_oracle_and_the_unfiltered_path(oracle):
    from trgt_amd import _lib
    rng = np.random.default_rng(31)
    plain_ctx = _lib.context_with_env(TRGT_NO_LONG_FILTER=1)
    loci = []
    for li in range(6):
        lf, rf = rand_dna(rng, 250), rand_dna(rng, 250)
        tr = (b"CAG" * 700)[:int(rng.integers(1500, 2100))]
        reads = []
        # flanks with more and more errors: the acceptance threshold (175 matches of 250) is crossed inside this series
        for e in (0, 10, 40, 60, 70, 74, 75, 76, 80, 90, 120):
            reads.append(rand_dna(rng, 900) + _mutate(rng, lf, n_sub=e) + tr + _mutate(rng, rf, n_sub=e // 2, n_del=e // 8, n_ins=e // 8) + rand_dna(rng, 700))
        # the left flank at chosen text positions: around the window borders (multiples of STEP, and the last window flush with the end)
        for pos in (STEP - 260, STEP - 250, STEP - 3, STEP, STEP + 1, 2 * STEP - 130, WL - 250, WL - 249, WL - 1, WL, 3 * STEP + 7):
            head = rand_dna(rng, pos)
            reads.append(head + _mutate(rng, lf, n_sub=int(rng.integers(1, 30))) + tr + _mutate(rng, rf, n_sub=3) + rand_dna(rng, 400))
        # a flank split by an insertion (two halves 60 / 200 / 400 bases apart), a flank cut by the end of the read, no flank at all,
        # two copies of the left flank (the worse one first)
        for gap in (60, 200, 400):
            reads.append(rand_dna(rng, 500) + lf[:125] + rand_dna(rng, gap) + lf[125:] + tr + rf + rand_dna(rng, 500))
        reads.append(lf[100:] + tr + rf + rand_dna(rng, 600))
        reads.append(rand_dna(rng, 800) + lf + tr + rf[:140])
        reads.append(rand_dna(rng, 3000 + 100 * li))
        reads.append(rand_dna(rng, 300) + _mutate(rng, lf, n_sub=50) + rand_dna(rng, 900) + _mutate(rng, lf, n_sub=2) + tr + _mutate(rng, rf, n_sub=1) + rand_dna(rng, 300))
        # short reads of the same locus go their usual way next to the long ones
        reads.append(rand_dna(rng, 100) + lf + b"CAG" * 10 + rf + rand_dna(rng, 100))
        reads.append(rand_dna(rng, 100) + _mutate(rng, lf, n_sub=5) + b"CAG" * 12 + rf[:200])
        loci.append(dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 20, motifs=[b"CAG"], ploidy=2, reads=reads))
    assert max(len(r) for L in loci for r in L["reads"]) > 4000
    n = _spans_equal(oracle, loci, plain_ctx)
    assert n >= 6 * 12  # (most of the series are located: the test is not vacuous)
    plain_ctx.close()


def test_many_long_reads_of_random_quality(oracle):
    from trgt_amd import _lib
    rng = np.random.default_rng(5)
    plain_ctx = _lib.context_with_env(TRGT_NO_LONG_FILTER=1)
    loci = []
    for li in range(10):
        lf, rf = rand_dna(rng, 250), rand_dna(rng, 250)
        reads = []
        for _ in range(20):
            tr = (b"GGCCCC" * 1700)[:int(rng.integers(800, 9000))]
            l = _mutate(rng, lf, n_sub=int(rng.integers(0, 100)), n_del=int(rng.integers(0, 12)), n_ins=int(rng.integers(0, 12)))
            r = _mutate(rng, rf, n_sub=int(rng.integers(0, 100)), n_del=int(rng.integers(0, 12)), n_ins=int(rng.integers(0, 12)))
            read = rand_dna(rng, int(rng.integers(0, 1500))) + l + tr + r + rand_dna(rng, int(rng.integers(0, 1500)))
            if rng.random() < 0.3:
                cut = int(rng.integers(300, len(read)))
                read = read[:cut] if rng.random() < 0.5 else read[-cut:]
            reads.append(read)
        loci.append(dict(left_flank=lf, right_flank=rf, tr=b"GGCCCC" * 10, motifs=[b"GGCCCC"], ploidy=2, reads=reads))
    _spans_equal(oracle, loci, plain_ctx)
    plain_ctx.close()
