"""Seeded windows of the flank fallback alignments (trgt_amd/csrc/spans.hip: piece_window, flank_window_kernel, window_check_kernel):
an implementation shortcut that must never show in the results.  Hand-made reads aim at its edges -- penalties right at the bound the
argument covers (11 for TRGT's penalties), every segment but one spoiled, flanks at the very ends of a read, flanks that occur twice,
periodic flanks (seeds on many diagonals), a better copy of the flank outside the seeded window -- and every read is compared with
the oracle's find_tr_spans / analyze_tr."""
import numpy as np
import pytest

from helpers import rand_dna

pytestmark = pytest.mark.gpu


def _sub(rng, seq, positions):
    b = bytearray(seq)
    for p in positions:
        b[p] = int(rng.choice([c for c in b"ACGT" if c != b[p]]))
    return bytes(b)


def _locus(rng, reads_fn, n_reads=12, periodic=None):
    if periodic:
        unit = rand_dna(rng, periodic)
        lf = (unit * (250 // periodic + 1))[:250]
        rf = rand_dna(rng, 250)
    else:
        lf, rf = rand_dna(rng, 250), rand_dna(rng, 250)
    tr = b"CAG" * 20
    reads = []
    for i in range(n_reads):
        reads.append(reads_fn(rng, i, lf, rf, tr))
    # two plain long reads keep the locus' longest read (and with it the short-read cut-off) where the others are "light"
    # (1220 bases: just below the length from which reads take the generic kernel, which has no windows)
    reads.append(rand_dna(rng, 330) + lf + tr + rf + rand_dna(rng, 330))
    reads.append(rand_dna(rng, 320) + lf + tr + rf + rand_dna(rng, 325))
    assert max(len(r) for r in reads) <= 1230 and min(len(r) for r in reads[:n_reads]) >= 925, sorted(len(r) for r in reads)
    return dict(left_flank=lf, right_flank=rf, tr=tr, motifs=[b"CAG"], ploidy=2, reads=reads)


def _check(oracle, loci, ctx=None):
    import torch
    from trgt_amd import locus
    from test_locus_gpu import _compare
    b = locus.pack(loci)
    p = locus.Params()
    outs = [("host reads", locus.run_batch(b, p, ctx=ctx)),
            ("device", locus.run_batch(b, p, ctx=ctx, flank_dev=torch.from_numpy(b["flank_blob"]).cuda(), reads_dev=torch.from_numpy(b["read_blob"]).cuda()))]
    for name, out in outs:
        _compare(oracle, locus, b, out, p, range(len(loci)))
    return b, outs[1][1]


def test_penalties_around_the_bound(oracle):
    rng = np.random.default_rng(2025)

    def reads_fn(rng, i, lf, rf, tr):
        # i mismatches in the left flank (penalty 2 i: 0..22), spread so that they hit different segments first
        pos = [(41 * j + 7 * (j // 6)) % 250 for j in range(i)]
        l = _sub(rng, lf, pos)
        # right flank: a deletion of i + 1 bases (penalty 5 + i + 1) plus, for odd i, one mismatch
        r = rf[:100] + rf[100 + i + 1:]
        if i % 2:
            r = _sub(rng, r, [30])
        return rand_dna(rng, 300) + l + tr + r + rand_dna(rng, 320)

    _check(oracle, [_locus(rng, reads_fn, n_reads=12) for _ in range(6)])


def test_insertions_and_all_but_one_segment_spoiled(oracle):
    rng = np.random.default_rng(7)

    def reads_fn(rng, i, lf, rf, tr):
        q = 250 // 6
        keep = i % 6                                     # the one segment left intact
        l = _sub(rng, lf, [s * q + 3 + i for s in range(6) if s != keep])          # five mismatches: penalty 10
        ins = rand_dna(rng, 1 + i % 7)                   # insertion of 1..7 bases: penalty 6..12
        r = rf[:125] + ins + rf[125:]
        return rand_dna(rng, 250 + 5 * i) + l + tr + r + rand_dna(rng, 300)

    _check(oracle, [_locus(rng, reads_fn, n_reads=14) for _ in range(5)])


def test_flanks_at_the_ends_of_the_read(oracle):
    rng = np.random.default_rng(11)

    def reads_fn(rng, i, lf, rf, tr):
        l = _sub(rng, lf, [17, 200][:1 + i % 2])
        r = _sub(rng, rf, [5, 90, 249][:1 + i % 3])
        left = rand_dna(rng, [0, 1, 3, 17, 40, 41, 42, 60][i % 8])        # window clipped at the start of the read
        right = rand_dna(rng, [0, 2, 16, 17, 18, 39, 45, 70][(i // 2) % 8])  # ... and at its end
        return left + l + tr * 9 + r + right           # (long repeat: the read stays above the short-read cut-off)

    _check(oracle, [_locus(rng, reads_fn, n_reads=16) for _ in range(4)])


def test_flank_twice_and_better_copy_elsewhere(oracle):
    rng = np.random.default_rng(13)

    def reads_fn(rng, i, lf, rf, tr):
        l1 = _sub(rng, lf, [20, 140])                    # penalty 4 ...
        l2 = _sub(rng, lf, [60])                         # ... and a better copy (penalty 2) further to the right / left
        gap = rand_dna(rng, [5, 30, 90, 200][i % 4])
        first, second = (l1, l2) if i % 2 else (l2, l1)
        r = _sub(rng, rf, [77])
        dup_r = rand_dna(rng, 40) + _sub(rng, rf, [10, 11, 12]) if i % 4 == 0 else b""   # a worse copy of the right flank behind it
        return rand_dna(rng, 45) + first + gap + second + tr + r + dup_r + rand_dna(rng, 45)

    _check(oracle, [_locus(rng, reads_fn, n_reads=12) for _ in range(5)])


@pytest.mark.parametrize("period", [1, 2, 3, 5, 8, 13, 41, 42, 83])
def test_periodic_flanks_seed_many_diagonals(oracle, period):
    rng = np.random.default_rng(100 + period)

    def reads_fn(rng, i, lf, rf, tr):
        l = _sub(rng, lf, [9 + 3 * i, 120][:1 + i % 2])
        if i % 4 == 3:
            l = l[:60] + l[60 + period:]                 # one period deleted: many equally good placements
        r = _sub(rng, rf, [50])
        return rand_dna(rng, 290) + l + tr + r + rand_dna(rng, 310)

    _check(oracle, [_locus(rng, reads_fn, n_reads=10, periodic=period) for _ in range(3)])


def test_windows_are_in_use_and_some_do_not_stand(oracle, capfd, monkeypatch):
    # the debug line of find_spans_device reports how many alignments ran on a window and how many of those were redone
    rng = np.random.default_rng(5)

    def reads_fn(rng, i, lf, rf, tr):
        l = _sub(rng, lf, [10 + 23 * j for j in range(i % 10)])    # 0..9 mismatches: penalties 0..18 (the argument covers up to 15)
        return rand_dna(rng, 300) + l + tr + _sub(rng, rf, [100]) + rand_dna(rng, 300)

    from trgt_amd import _lib
    dctx = _lib.context_with_env(TRGT_WFA_DEBUG=1)  # (planner knobs are read when a context is created)
    try:
        _check(oracle, [_locus(rng, reads_fn, n_reads=16) for _ in range(4)], ctx=dctx)
    finally:
        dctx.close()
    err = capfd.readouterr().err
    line = [l for l in err.splitlines() if l.startswith("[spans]")][-1]
    nums = [int(t) for t in line.replace(",", " ").replace("(", " ").replace(")", " ").split() if t.isdigit()]
    windowed, redone, shortcut = nums[3], nums[-3], nums[-2]  # (the last number: how many of the shortcut's were one-base gaps)
    assert windowed > 30 and redone >= 4 and shortcut >= 64, line  # (every right flank and the left flanks with 1-2 mismatches need no alignment)


@pytest.mark.parametrize("scoring,flank_len", [((1, 2, 1), 250), ((4, 6, 2), 250), ((2, 5, 1), 150), ((2, 5, 1), 100), ((3, 1, 1), 250), ((1, 0, 1), 200)])
def test_other_penalties_and_flank_lengths(oracle, scoring, flank_len):
    # the bound, the margins and whether windows are used at all follow from the penalties and the flank length (window_plan in
    # find_spans_device): synthetic loci under other settings, every locus against the oracle
    import torch
    from trgt_amd import locus, synth
    from test_locus_gpu import _compare
    b = synth.generate(48, first_locus=4242 + flank_len)
    p = locus.Params(aln_scoring=scoring, search_flank_len=flank_len)
    out = locus.run_batch(b, p, flank_dev=torch.from_numpy(b["flank_blob"]).cuda(), reads_dev=torch.from_numpy(b["read_blob"]).cuda())
    _compare(oracle, locus, b, out, p, range(48))
    # the pre-filter is engaged for the two presets that have an instantiation (2,5,1 and the targeted 1,0,1): it counts its offsets
    engaged = int(out.stats[17]) > 0
    assert engaged == (tuple(scoring) in ((2, 5, 1), (1, 0, 1))), (scoring, int(out.stats[17]))


def test_substitution_only_shortcut(oracle):
    """One or two substitutions in a flank, all seeds on one diagonal: the window search settles the alignment itself (stats[21]).  Aimed
    at its edges: mismatches in the first / last bases and in the two bytes after the last full dword, three mismatches (not covered),
    a second copy of the flank with more mismatches elsewhere in the read (seeds on two diagonals: not covered), flanks flush with the
    ends of the read, a mismatch next to a deletion."""
    rng = np.random.default_rng(99)

    def reads_fn(rng, i, lf, rf, tr):
        sets = [[0], [249], [248], [247, 249], [0, 1], [124], [30, 200], [3, 100, 220], [60], [10, 11], [246], [125, 126]]
        l = _sub(rng, lf, sets[i % len(sets)])
        r = _sub(rng, rf, sets[(i + 5) % len(sets)])
        head, tail = rand_dna(rng, 300 + i), rand_dna(rng, 310)
        if i == 4:    # a worse copy of the left flank further left (four mismatches): seeds on two diagonals
            head = rand_dna(rng, 20) + _sub(rng, lf, [5, 70, 140, 210]) + rand_dna(rng, 40)
        if i == 6:    # flank flush with the start of the read
            head = b""
            tail = rand_dna(rng, 600)
        if i == 8:    # ... and with its end
            tail = b""
            head = rand_dna(rng, 620)
        if i == 9:    # a substitution and a deletion in the same flank: not substitution-only
            r = r[:180] + r[181:]
        return head + l + tr + r + tail

    import torch
    from trgt_amd import _lib, locus
    loci = [_locus(rng, reads_fn, n_reads=12) for _ in range(5)]
    b, out = _check(oracle, loci)
    assert int(out.stats[21]) >= 5 * 12  # most of the 2 x 12 flank pieces per locus are settled without an alignment
    import os
    os.environ["TRGT_NO_HAMMING"] = "1"
    try:
        ctx = _lib.Context(0)
        _, out2 = _check(oracle, loci, ctx=ctx)
    finally:
        del os.environ["TRGT_NO_HAMMING"]
    assert int(out2.stats[21]) == 0
    for k in ("span_start", "span_end", "allele_len", "classification"):
        assert np.array_equal(getattr(out, k), getattr(out2, k)), k


def test_one_base_gap_shortcut(oracle):
    """One inserted or deleted base in a flank, seeds on two neighbouring diagonals: settled by the window search without an alignment.
    Gaps at every kind of position -- near the ends (inside the margins: left to the aligner), in the middle, inside homopolymer and
    dinucleotide runs (the place of the gap is open, the result is not), next to a substitution (not covered), two gaps (not covered) --
    and flanks of low complexity, where prefix and suffix match on both diagonals."""
    rng = np.random.default_rng(4242)

    def with_run(rng, flank, at, unit, n):
        b = bytearray(flank)
        run = (unit * (n // len(unit) + 1))[:n]
        b[at:at + n] = run
        return bytes(b)

    def reads_fn(rng, i, lf, rf, tr):
        pos = [1, 5, 12, 14, 20, 31, 62, 100, 125, 180, 214, 216, 218, 230, 247, 249]
        p = pos[i % len(pos)]
        l, r = lf, rf
        if i % 3 == 0:
            l = l[:p] + rand_dna(rng, 1) + l[p:]          # inserted base (may equal a neighbour: then the place is open)
        elif i % 3 == 1:
            l = l[:p] + l[p + 1:]                          # deleted base
        else:
            l = l[:p] + l[p + 1:]
            l = _sub(rng, l, [(p + 40) % 240])             # ... and a substitution: the aligner's job
        q = pos[(i + 7) % len(pos)]
        if i % 2:
            r = r[:q] + r[q:q + 1] + r[q:]                 # a base doubled (an insertion inside a run of at least two)
        else:
            r = r[:q] + r[q + 1:]
        if i == 5:                                         # two gaps in one flank
            r = rf[:60] + rf[61:150] + b"A" + rf[150:]
        return rand_dna(rng, 300 + i) + l + tr + r + rand_dna(rng, 310)

    loci = [_locus(rng, reads_fn, n_reads=16) for _ in range(4)]
    # flanks with long runs: a 40-base homopolymer and a dinucleotide run in the left flank, gaps inside and next to them
    for unit, at in ((b"A", 90), (b"AC", 120), (b"T", 30)):
        base = dict(left_flank=rand_dna(rng, 250), right_flank=rand_dna(rng, 250), tr=b"CAG" * 20)
        lf = with_run(rng, base["left_flank"], at, unit, 40)
        rf = base["right_flank"]
        reads = []
        for i in range(16):
            p = at - 3 + 3 * i
            l = lf[:p] + lf[p + 1:] if i % 2 else lf[:p] + lf[p:p + 1] + lf[p:]
            reads.append(rand_dna(rng, 305) + l + base["tr"] + rf + rand_dna(rng, 300))
        reads += [rand_dna(rng, 330) + lf + base["tr"] + rf + rand_dna(rng, 330), rand_dna(rng, 320) + lf + base["tr"] + rf + rand_dna(rng, 325)]
        loci.append(dict(left_flank=lf, right_flank=rf, tr=base["tr"], motifs=[b"CAG"], ploidy=2, reads=reads))
    import os
    from trgt_amd import _lib
    b, out = _check(oracle, loci)
    os.environ["TRGT_NO_INDEL_SHORTCUT"] = "1"
    try:
        ctx = _lib.Context(0)
        _, out2 = _check(oracle, loci, ctx=ctx)
    finally:
        del os.environ["TRGT_NO_INDEL_SHORTCUT"]
    assert int(out.stats[21]) >= int(out2.stats[21]) + 40  # the gaps outside the margins are settled without an alignment
    for k in ("span_start", "span_end", "allele_len", "classification"):
        assert np.array_equal(getattr(out, k), getattr(out2, k)), k


def _short_read_locus(rng, reads_fn, n_reads):
    # as _locus, but the reads under test are at least 300 bases shorter than the locus' longest read: "too short to span the locus",
    # the list that meets the pre-filter, and what it keeps is back-traced inside a band (heavy_band_kernel / band_check_kernel)
    lf, rf = rand_dna(rng, 250), rand_dna(rng, 250)
    tr = b"CAG" * 20
    reads = [reads_fn(rng, i, lf, rf, tr) for i in range(n_reads)]
    reads.append(rand_dna(rng, 330) + lf + tr + rf + rand_dna(rng, 330))
    reads.append(rand_dna(rng, 320) + lf + tr + rf + rand_dna(rng, 325))
    assert max(len(r) for r in reads[:n_reads]) < 915 and min(len(r) for r in reads[:n_reads]) >= 250, sorted(len(r) for r in reads)
    return dict(left_flank=lf, right_flank=rf, tr=tr, motifs=[b"CAG"], ploidy=2, reads=reads)


def _no_seed(rng, flank, extra=0):
    # a mismatch inside the first twelve bases of each of the eight segments (no seed anywhere: penalty 16), then `extra` more
    pos = [31 * s + 2 + (5 * s) % 9 for s in range(8)] + [31 * (j % 8) + 14 + j // 8 * 3 for j in range(extra)]
    return _sub(rng, flank, pos)


@pytest.mark.parametrize("band", [None, "0", "24", "256"])
def test_kept_alignments_run_inside_their_band(oracle, capfd, band):
    """What the pre-filter keeps is aligned again with a back-trace, inside the diagonals its penalty and end diagonal allow.  Reads too
    short to span the locus whose left flank has no seed: penalties from 16 to beyond the default band (96), insertions and deletions
    (the band must hold the whole path, not just its end), the flank at the very start / end of the read (band clipped by the text), the
    flank twice at the same penalty (the first end diagonal wins) and reads without it.  Same results as the oracle whatever the band."""
    rng = np.random.default_rng(4242)

    def reads_fn(rng, i, lf, rf, tr):
        k = i % 12
        l = _no_seed(rng, lf, extra=(0, 3, 8, 4, 0, 2, 0, 0, 1, 0, 0, 6)[k])
        if k == 3:    # a 20-base insertion and a 6-base deletion in the flank: penalty 16 + 8 + 25 + 11
            l = l[:90] + rand_dna(rng, 20) + l[90:170] + l[176:]
        elif k == 4:  # three long insertions: beyond the default band
            l = l[:60] + rand_dna(rng, 30) + l[60:130] + rand_dna(rng, 30) + l[130:200] + rand_dna(rng, 28) + l[200:]
        elif k == 5:  # two insertions: close to the default band
            l = l[:80] + rand_dna(rng, 28) + l[80:180] + rand_dna(rng, 30) + l[180:]
        head = rand_dna(rng, 0 if k == 6 else 3 if k == 7 else int(rng.integers(20, 200)))
        if k == 8:    # the same damaged flank twice
            head = head + l + rand_dna(rng, 40)
        if k == 9:    # no left flank at all
            l = rand_dna(rng, 250)
        r = _sub(rng, rf, [100]) if k != 10 else b""  # k == 10: the read ends with the left flank
        tail = rand_dna(rng, int(rng.integers(5, 60))) if k != 10 else b""
        mid = tr if k != 10 else b""
        return (head + l + mid + r + tail)[:900]

    from trgt_amd import _lib
    env = dict(TRGT_WFA_DEBUG=1)
    if band is not None:
        env["TRGT_HEAVY_BAND"] = band
    dctx = _lib.context_with_env(**env)
    try:
        _check(oracle, [_short_read_locus(rng, reads_fn, n_reads=24) for _ in range(5)], ctx=dctx)
    finally:
        dctx.close()
    lines = [l for l in capfd.readouterr().err.splitlines() if l.startswith("[spans+] kept by the pre-filter")]
    if band == "0":
        assert not lines
        return
    nums = [int(t) for t in lines[-1].replace(",", " ").replace("(", " ").replace(")", " ").replace(":", " ").split() if t.isdigit()]
    kept, banded, failed, whole = nums
    assert failed == 0 and kept == banded + whole and kept >= 60, lines[-1]
    if band is None:
        assert banded >= 50 and whole >= 5, lines[-1]   # (the three long insertions: penalty above 96)
    if band == "24":
        assert 10 <= banded < kept - 20, lines[-1]
    if band == "256":
        assert whole == 0, lines[-1]
