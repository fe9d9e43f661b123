"""Parity of the HIP motif-HMM path (trgt_hmm_batch through the C ABI) against the CPU oracle and the
reference's own known-answer tests.  Bit-exact: state paths, spans, counts, and the f64 purity bits."""
import json
import math
import os

import numpy as np
import pytest

from helpers import rand_dna, rand_motif, repeat_allele

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hmm_kats.json")))["kats"]


@pytest.fixture(scope="module")
def hmm():
    from trgt_amd import hmm as H
    return H


def _same(oracle, H, motif_sets, jobs, want_path=True):
    batch = H.pack_hmm_batch(motif_sets, jobs)
    got = H.hmm_batch(batch, want_path=want_path)
    ref = oracle.hmm_batch(batch, n_threads=4, want_path=want_path)
    assert np.array_equal(got["n_spans"], ref["n_spans"])
    assert np.array_equal(got["path_len"], ref["path_len"])
    assert np.array_equal(got["edit"], ref["edit"]) and np.array_equal(got["maxd"], ref["maxd"])
    assert np.array_equal(got["purity"].view(np.uint64), ref["purity"].view(np.uint64))  # bit-exact incl. NaN payload sign
    assert np.array_equal(got["counts"], ref["counts"])
    for j in range(len(jobs)):
        so, ns = int(batch["span_off"][j]), int(ref["n_spans"][j])
        assert np.array_equal(got["spans"][3 * so:3 * (so + ns)], ref["spans"][3 * so:3 * (so + ns)]), j
        if want_path:
            po, pl = int(batch["path_off"][j]), int(ref["path_len"][j])
            assert np.array_equal(got["path"][po:po + pl], ref["path"][po:po + pl]), j
    return got


def test_reference_kats_through_abi(oracle, hmm):
    for kat in KATS:
        if "summary" in kat and not kat["remove_imperfect"]:
            path = hmm.build_hmm(kat["motifs"]).label(kat["query"])
            sp = oracle.hmm_label_motifs(kat["motifs"], path)
            out = []
            for m, s, e in sp.tolist():
                if out and out[-1][2] == m:
                    out[-1][1] = e
                else:
                    out.append([s, e, m])
            assert out == kat["summary"], kat["id"]
        elif "purity" in kat:
            a = hmm.label_with_hmm(kat["motifs"], [kat["query"]])[0]
            if kat["purity"] is None:
                assert math.isnan(a.purity) and a.labels is None
            else:
                assert a.purity == kat["purity"][0] / kat["purity"][1], kat["id"]


def test_kat_spans_after_remove_imperfect(hmm):
    # builder.rs:218-240 / 252-273 via the fused label_with_hmm path: skip spans dropped, spans collapsed
    for kat in KATS:
        if kat.get("remove_imperfect"):
            a = hmm.label_with_hmm(kat["motifs"], [kat["query"]])[0]
            want = [[s, e, m] for s, e, m in kat["summary"] if m < len(kat["motifs"])]
            got = [[s.start, s.end, s.motif_index] for s in (a.labels or [])]
            merged = []
            for s, e, m in want:  # collapse_labels merges abutting same-motif spans
                if merged and merged[-1][2] == m and merged[-1][1] == s:
                    merged[-1][1] = e
                else:
                    merged.append([s, e, m])
            assert got == merged, kat["id"]


def test_random_single_motif_str(oracle, hmm):
    rng = np.random.default_rng(20250509)
    sets, jobs = [], []
    for s in range(300):
        sets.append([rand_motif(rng, 3, 6, allow_n=False)])
        for _ in range(2):
            jobs.append((s, repeat_allele(rng, sets[-1], int(rng.integers(1, 201)), err=0.01)))
    _same(oracle, hmm, sets, jobs)


def test_random_multi_motif_and_n(oracle, hmm):
    rng = np.random.default_rng(7)
    sets, jobs = [], []
    for s in range(120):
        sets.append([rand_motif(rng, 1, 12) for _ in range(int(rng.integers(1, 6)))])
        for _ in range(2):
            jobs.append((s, repeat_allele(rng, sets[-1], int(rng.integers(0, 400)), err=0.03)))
    _same(oracle, hmm, sets, jobs)


def test_edge_cases(oracle, hmm):
    sets = [[b"A"], [b"N"], [b"CAG", b"CCG"], [b"GCN"], [b"AC"], [b"ACGTACGTACGTACGTACGTAAAA"]]
    jobs = [(0, b""), (0, b"A"), (0, b"AAAAAAAAAA"), (0, b"CCCC"), (1, b"ACGTACGT"), (2, b""), (2, b"C"),
            (2, b"CAGCAGCAGTTTTTTTTCCGCCGCCG"), (3, b"GCAGCCGCTGAG"), (4, b"ACACACACAC"), (4, b"CACACA"), (4, b"T"),
            (5, b"ACGTACGTACGTACGTACGTAAAA" * 3), (2, b"CAGNNNCAGRYCAG"), (0, b"NNNN")]
    _same(oracle, hmm, sets, jobs)


def test_many_motifs_multi_wave_workgroup(oracle, hmm):
    # S = 170 (RFC1-like 10-motif set) -> 192-thread workgroups
    rng = np.random.default_rng(3)
    sets = [[b"AAAAG", b"AAAGG", b"AAGGG", b"AAGAG", b"AGAGG", b"AACGG", b"GGGAC", b"AAAGGG", b"AAAAGG", b"AAGAC"]]
    assert hmm.num_states(sets[0]) > 128
    jobs = [(0, repeat_allele(rng, sets[0], n, err=0.02)) for n in (5, 60, 333, 1000)]
    base = _same(oracle, hmm, sets, jobs)
    from trgt_amd import _lib
    ctx = _lib.context_with_env(TRGT_HMM_FOUR_ROUNDS=1)  # four barriers per column (the loop the two-barrier fill falls back to)
    try:
        a = hmm.hmm_batch(hmm.pack_hmm_batch(sets, jobs), ctx=ctx)
    finally:
        ctx.close()
    for k in ("n_spans", "path_len", "edit", "maxd", "counts", "spans"):
        assert np.array_equal(a[k], base[k]), k
    assert np.array_equal(a["purity"].view(np.uint64), base["purity"].view(np.uint64))


def test_long_allele_10kb(oracle, hmm):
    rng = np.random.default_rng(11)
    sets = [[b"CAG"], [b"GGCCTG", b"CCG"]]
    jobs = [(0, repeat_allele(rng, sets[0], 10000, err=0.005)), (1, repeat_allele(rng, sets[1], 6000, err=0.01))]
    _same(oracle, hmm, sets, jobs)


def test_long_alleles_parallel_traceback(oracle, hmm):
    # alleles of 1 536 columns and more are traced back by hmm_traceback_long_kernel (chunk maps on many waves, then every chunk again
    # from its known entry state): lengths around the threshold and around chunk boundaries (64 columns), one-wave, two-alleles-per-wave
    # and multi-wave models, one-base motifs (a motif visit per column), with and without state paths -- and the same batch with the
    # kernel switched off (TRGT_HMM_NO_LONG_TB=1)
    from trgt_amd import _lib
    rng = np.random.default_rng(29)
    sets = [[b"CAG"], [b"A"], [b"GGCCTG", b"CCG"], [b"AAAAG", b"AAAGG", b"AAGGG", b"AAGAG", b"AGAGG", b"AACGG", b"GGGAC", b"AAAGGG", b"AAAAGG", b"AAGAC"],
            [b"ACGTTGCAAGGCTTAACCGTAC"]]
    jobs = []
    for n in (1533, 1534, 1535, 1598, 1599, 1600, 1662, 4000):
        jobs.append((0, repeat_allele(rng, sets[0], n, err=0.02)))
    for n in (1534, 2047, 3000):
        jobs.append((1, b"A" * n))
        jobs.append((2, repeat_allele(rng, sets[2], n, err=0.03)))
    for n in (1540, 2600):
        jobs.append((3, repeat_allele(rng, sets[3], n, err=0.02)))
        jobs.append((4, repeat_allele(rng, sets[4], n, err=0.02)))
    jobs.append((0, rand_dna(rng, 2000)))  # nothing of the motif: skip states all the way
    for want_path in (True, False):
        _same(oracle, hmm, sets, jobs, want_path=want_path)
    batch = hmm.pack_hmm_batch(sets, jobs)
    b = hmm.hmm_batch(batch)
    # ... switched off; by one workgroup per allele in one launch; by three and by sixteen workgroups per allele (default: four)
    for env in (dict(TRGT_HMM_NO_LONG_TB=1), dict(TRGT_HMM_LONG_WGS=1), dict(TRGT_HMM_LONG_WGS=3), dict(TRGT_HMM_LONG_WGS=16)):
        ctx = _lib.context_with_env(**env)
        try:
            a = hmm.hmm_batch(batch, ctx=ctx)
        finally:
            ctx.close()
        for k in ("n_spans", "path_len", "edit", "maxd", "counts", "spans"):
            assert np.array_equal(a[k], b[k]), (env, k)
        assert np.array_equal(a["purity"].view(np.uint64), b["purity"].view(np.uint64)), env
        for j in range(len(jobs)):  # (behind a job's path the buffer holds what the reversal left there)
            po, pl = int(batch["path_off"][j]), int(a["path_len"][j])
            assert np.array_equal(a["path"][po:po + pl], b["path"][po:po + pl]), (env, j)


def test_register_fill_variants_agree(oracle, hmm):
    # one-wave models fill their columns in registers with two rounds of cross-lane fetches per column (run end, run start and block
    # starts worked out by every lane from one round); TRGT_HMM_FOUR_ROUNDS=1 takes the loop with one round per pass, TRGT_HMM_LDS_FILL=1
    # the LDS columns of the larger models (two barriers per column; with TRGT_HMM_FOUR_ROUNDS=1 four).  One-motif models (two alleles per wave), models up to 64 states, motifs of one and two bases
    # (no deletion states / one), N in motifs and alleles, alleles without the motif, empty alleles -- all three like the oracle.
    from trgt_amd import _lib
    rng = np.random.default_rng(911)
    sets = [[b"CAG"], [b"A"], [b"AC"], [b"GCN"], [b"CAG", b"CCG"], [b"A", b"T", b"CG"], [b"AAGGG", b"AAAAG", b"ACG"], [b"ACGTTGCA"],
            [b"GGCCTG", b"CCG", b"A"], [b"ACGTACGTACGTACGTAC"]]
    assert max(hmm.num_states(m) for m in sets) <= 64 and min(hmm.num_states(m) for m in sets) <= 11
    jobs = []
    for s, m in enumerate(sets):
        for n in (0, 1, 2, 7, 40, 150, 333):
            jobs.append((s, repeat_allele(rng, m, n, err=0.04)))
        jobs.append((s, rand_dna(rng, 120)))
        jobs.append((s, b"N" * 9 + repeat_allele(rng, m, 30, err=0.0) + b"NN"))
    base = _same(oracle, hmm, sets, jobs)
    batch = hmm.pack_hmm_batch(sets, jobs)
    # (round 5: by default the fill runs with one lane per motif position, hmm_ppl.hpp, in front of those kernels; TRGT_HMM_NO_PPL=1 takes
    #  the fills of hmm_viterbi_kernel, which the other switches select among)
    for env in (dict(TRGT_HMM_NO_PPL=1), dict(TRGT_HMM_NO_PPL=1, TRGT_HMM_FOUR_ROUNDS=1), dict(TRGT_HMM_NO_PPL=1, TRGT_HMM_LDS_FILL=1),
                dict(TRGT_HMM_NO_PPL=1, TRGT_HMM_LDS_FILL=1, TRGT_HMM_FOUR_ROUNDS=1),
                dict(TRGT_HMM_PPL_WIDE=1)):  # (round 6: the position-per-lane fill's rows of one byte per state; default: one byte per lane)
        ctx = _lib.context_with_env(**env)
        try:
            a = hmm.hmm_batch(batch, ctx=ctx)
        finally:
            ctx.close()
        for k in ("n_spans", "path_len", "edit", "maxd", "counts", "spans"):
            assert np.array_equal(a[k], base[k]), (env, k)
        assert np.array_equal(a["purity"].view(np.uint64), base["purity"].view(np.uint64)), env
        for j in range(len(jobs)):
            po, pl = int(batch["path_off"][j]), int(a["path_len"][j])
            assert np.array_equal(a["path"][po:po + pl], base["path"][po:po + pl]), (env, j)


def test_position_per_lane_fill_against_the_state_fill_on_long_alleles(oracle, hmm):
    # the position-per-lane fill writes SOME valid predecessor slot for states without a score where the fill of
    # hmm_viterbi_kernel writes 0xFF; every consumer must ignore those bytes.  Alleles long enough for the staged trace-back (>= 512
    # columns) and for the chunk-map path (>= 1 536 columns: hmm_traceback_long_kernel), small and large motif sets, with errors (states
    # off the path carry no score): both fills must give the same paths, spans, counts and purity bits -- and the oracle's.
    from trgt_amd import _lib
    rng = np.random.default_rng(4242)
    sets = [[b"CAG"], [b"AAGGG", b"AAAAG"], [b"GGCCTG", b"CCG", b"A"], [b"ACGTACGTACGTACGTAC"],
            [b"AAAAG", b"AAAGG", b"AAGGG", b"AAGAG", b"AGAGG", b"AACGG", b"GGGAC", b"AAAGGG", b"AAAAGG", b"AAGAC"]]
    jobs = []
    for s, m in enumerate(sets):
        for n_bases in (520, 700, 1540, 2100, 3300):
            jobs.append((s, repeat_allele(rng, m, n_bases, err=0.03)))
    base = _same(oracle, hmm, sets, jobs)
    assert max(len(a) for _, a in jobs) >= 1536 and min(len(a) for _, a in jobs) >= 400
    batch = hmm.pack_hmm_batch(sets, jobs)
    for env in (dict(TRGT_HMM_NO_PPL=1), dict(TRGT_HMM_NO_PPL=1, TRGT_HMM_NO_LONG_TB=1), dict(TRGT_HMM_NO_LONG_TB=1),
                dict(TRGT_HMM_PPL_WIDE=1), dict(TRGT_HMM_PPL_WIDE=1, TRGT_HMM_NO_LONG_TB=1)):
        ctx = _lib.context_with_env(**env)
        try:
            a = hmm.hmm_batch(batch, ctx=ctx)
        finally:
            ctx.close()
        for k in ("n_spans", "path_len", "edit", "maxd", "counts", "spans"):
            assert np.array_equal(a[k], base[k]), (env, k)
        assert np.array_equal(a["purity"].view(np.uint64), base["purity"].view(np.uint64)), env
        for j in range(len(jobs)):
            po, pl = int(batch["path_off"][j]), int(a["path_len"][j])
            assert np.array_equal(a["path"][po:po + pl], base["path"][po:po + pl]), (env, j)


def test_visit_list_overflow_and_length_buckets(oracle, hmm):
    # one-base motifs make one motif visit per base: more than the kernel keeps in LDS (the rest go through its global
    # workspace); alleles of 0..2500 bases of several models land in every length bucket / launch class of one batch
    rng = np.random.default_rng(17)
    sets = [[b"A"], [b"A", b"C"], [b"CAG"], [b"AAAAG", b"AAAGG", b"AAGGG", b"AAGAG", b"AGAGG", b"AACGG", b"GGGAC", b"AAAGGG", b"AAAAGG", b"AAGAC"]]
    jobs = []
    for n in (1, 63, 64, 65, 150, 192, 193, 400, 640, 641, 1200, 2048, 2049, 2500):
        jobs.append((0, b"A" * n))
        jobs.append((1, bytes(rng.choice(np.frombuffer(b"AACCCA", np.uint8), n).tobytes())))
        jobs.append((2, repeat_allele(rng, sets[2], n, err=0.02)))
    for n in (100, 300, 900):
        jobs.append((3, repeat_allele(rng, sets[3], n, err=0.02)))
    _same(oracle, hmm, sets, jobs)


def test_device_built_models_equal_host_builder(hmm):
    # the model tables (ln transition / emission terms, predecessor lists, block table, lane table of multi-wave models) are built
    # by a kernel; the host-side builder restates build_hmm / define_motif_block (builder.rs:4-173) independently: byte-identical
    rng = np.random.default_rng(23)
    sets = [[b"A"], [b"N"], [b"CAG", b"CCG"], [b"GCN"], [b"AC"], [b"ACGTACGTACGTACGTACGTAAAA"], [b"AXGT", b"R"],
            [b"AAAAG", b"AAAGG", b"AAGGG", b"AAGAG", b"AGAGG", b"AACGG", b"GGGAC", b"AAAGGG", b"AAAAGG", b"AAGAC"]]
    for _ in range(300):
        sets.append([rand_motif(rng, 1, 12) for _ in range(int(rng.integers(1, 5)))])
    for n in (20, 59, 60, 64, 65, 70, 130, 200):  # (the kernel takes models of up to 1024 states)
        sets.append([rand_dna(rng, n)])
        sets.append([rand_dna(rng, n), rand_dna(rng, 3), rand_dna(rng, n // 2 + 1)])
    sets.append([rand_dna(rng, 338)])
    assert hmm.models_check(sets) == 0


def test_device_resident_inputs(oracle, hmm):
    import torch
    rng = np.random.default_rng(5)
    sets = [[b"CAG"]]
    jobs = [(0, repeat_allele(rng, sets[0], 150)) for _ in range(64)]
    batch = hmm.pack_hmm_batch(sets, jobs)
    dev = torch.from_numpy(batch["seq_blob"]).cuda()
    got = hmm.hmm_batch(batch, want_path=False, seq_blob_dev=dev)
    ref = oracle.hmm_batch(batch, want_path=False)
    assert np.array_equal(got["purity"].view(np.uint64), ref["purity"].view(np.uint64))
    assert np.array_equal(got["counts"], ref["counts"])


def test_position_per_lane_fill_shapes(oracle, hmm):
    # hmm_fill_ppl_kernel: one lane per motif position, 8 / 16 / 32 / 64 lanes per allele.  Sets on every group width and at its edges
    # (positions + 1 = 8, 9, 16, 17, 32, 33, 64; 65 positions do not fit and take hmm_viterbi_kernel's own fill), one-base motifs (no
    # deletion states), many blocks (the run end's first-maximum over up to 21 block ends), long motifs (the deletion chain's early
    # exit), alleles of different lengths side by side in one wave, empty alleles, alleles across the 256-column code window.
    rng = np.random.default_rng(505)
    sets = [[b"CAG"], [b"ACGTTGC"], [b"ACGTTGCA"], [b"A"] * 1 + [b"C", b"G", b"T", b"AC", b"AG"], [b"ACGTA", b"CCGGA", b"TTGAC"], [b"ACGTACGTACGTTGCA"],
            [rand_dna(rng, 31)], [rand_dna(rng, 32)], [rand_dna(rng, 63)], [rand_dna(rng, 64)], [b"A", b"C", b"G", b"T"] * 5, [b"AAGGG", b"AAAAG", b"ACG", b"N", b"GCN"]]
    jobs = []
    for s, m in enumerate(sets):
        for n in (0, 1, 3, 20, 90):
            jobs.append((s, repeat_allele(rng, m, n, err=0.03)))
        jobs.append((s, rand_dna(rng, 300)))
    for n in (254, 255, 256, 257, 258, 511, 512, 513, 700):
        jobs.append((0, (b"CAG" * 300)[:n]))
    _same(oracle, hmm, sets, jobs)
