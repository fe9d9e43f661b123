"""CPU checks of the oracle's cluster genotyper pieces.

The Ward linkage lives in kodama 0.3.0, an un-vendored crates.io dependency of the reference (Cargo.lock:839-842), and the
reference has no test of genotype_cluster::cluster(): PARITY UNPINNED.  What can be checked here is that the oracle's
restatement of the NN-chain algorithm builds the same dendrogram as an independent implementation
(scipy.cluster.hierarchy.linkage(method="ward")) on tie-free inputs, and that cluster() / genotype() behave as
genotype_cluster.rs:58-227 reads.
"""
import numpy as np
import pytest
from scipy.cluster.hierarchy import linkage
from scipy.spatial.distance import pdist


@pytest.mark.parametrize("n,seed", [(3, 0), (5, 1), (12, 2), (30, 3), (64, 4), (250, 5)])
def test_ward_dendrogram_matches_scipy(oracle, n, seed):
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(n, 3)) + (rng.integers(0, 2, size=(n, 1)) * 6.0)  # two blobs, tie-free euclidean distances
    d = pdist(pts)
    steps, diss, mutated = oracle.ward_linkage(d, n)
    Z = linkage(d, method="ward")
    assert len(steps) == n - 1
    assert np.array_equal(steps[:, 0], Z[:, 0].astype(int)) and np.array_equal(steps[:, 1], Z[:, 1].astype(int))
    assert np.array_equal(steps[:, 2], Z[:, 3].astype(int))
    assert np.allclose(diss, Z[:, 2], rtol=1e-9, atol=1e-12)
    assert not np.array_equal(mutated, d)  # kodama squares and updates the caller's matrix in place


def test_ward_mutates_matrix_like_lance_williams(oracle):
    # 3 points: after merging the closest pair (a, b) the surviving entry holds the squared Ward distance to the third
    d = np.array([1.0, 4.0, 5.0])  # d01, d02, d12
    steps, diss, m = oracle.ward_linkage(d, 3)
    assert steps.tolist() == [[0, 1, 2], [2, 3, 3]]
    assert m[0] == 1.0 and m[1] == 16.0  # squared, rows of the absorbed cluster 0 untouched afterwards
    want = ((1 + 1) * 16.0 + (1 + 1) * 25.0 - 1 * 1.0) / 3.0
    assert m[2] == want and diss[1] == np.sqrt(want)


def test_cluster_groups_two_clear_alleles(oracle):
    # 6 reads of length 30 and 5 of length 60 (distance = sqrt(|len diff|) as get_dist falls back to for long pairs)
    lens = np.array([30] * 6 + [60] * 5)
    n = len(lens)
    d = np.array([np.sqrt(abs(int(lens[i]) - int(lens[j]))) for i in range(n) for j in range(i + 1, n)])
    k, g, _ = oracle.cluster_groups(d, n)
    assert k == 2
    assert len(set(g[:6])) == 1 and len(set(g[6:])) == 1 and g[0] != g[6]


def test_cluster_groups_homozygous_split_is_even_odd(oracle):
    n = 9
    d = np.zeros(n * (n - 1) // 2)
    k, g, _ = oracle.cluster_groups(d, n)  # identical reads: the chain absorbs one singleton per step, no balanced split
    assert k == 2 and g.tolist() == [i % 2 for i in range(n)]
    k, g, _ = oracle.cluster_groups(np.array([1.0]), 2)
    assert k == 2 and g.tolist() == [0, 1]
    # one far outlier among identical reads: no split has >= 2 members on both sides -> cutoff stays 0.0 -> even / odd
    n = 8
    d = np.array([3.0 if j == n - 1 else 0.0 for i in range(n) for j in range(i + 1, n)])
    k, g, _ = oracle.cluster_groups(d, n)
    assert k == 2 and g.tolist() == [i % 2 for i in range(n)]


def _locus(rng, alleles, n_reads, motif=b"CAG", err=0.0):
    from helpers import mutate, rand_dna
    lf, rf = rand_dna(rng, 250), rand_dna(rng, 250)
    lc, rc = rand_dna(rng, 250), rand_dna(rng, 250)
    reads = []
    for i in range(n_reads):
        a = alleles[i % len(alleles)]
        reads.append(mutate(rng, lc + lf + a + rf + rc, err, err / 2, err / 2) if err else lc + lf + a + rf + rc)
    return lf, rf, reads


def test_cluster_genotyper_two_alleles(oracle):
    rng = np.random.default_rng(7)
    a1, a2 = b"CAG" * 10, b"CAG" * 4 + b"CCG" * 9
    lf, rf, reads = _locus(rng, [a1, a2], 16)
    r = oracle.locus_analyze(lf, rf, a1, [b"CAG", b"CCG"], reads, genotyper=1)
    assert r["n_alleles"] == 2 and r["alleles"] == [a1.decode(), a2.decode()]
    assert r["stats"]["n_wfa_ed"] == 16 * 15 // 2 and r["stats"]["n_wfa_cons"] == 16
    assert r["classification"].tolist() == [0] * 8 + [1] * 8  # LocusResult.reads order: stable sort by span length
    assert r["kept_read"].tolist() == list(range(0, 16, 2)) + list(range(1, 16, 2))
    assert r["SD"] == "8,8" and r["AL"] == "30,39"
    # haploid: one consensus over all reads
    r1 = oracle.locus_analyze(lf, rf, a1, [b"CAG", b"CCG"], reads[::2], genotyper=1, ploidy=1)
    assert r1["n_alleles"] == 1 and r1["alleles"] == [a1.decode()] and r1["stats"]["n_wfa_cons"] == 8
    # a single read with ploidy 2 -> the same consensus twice
    r2 = oracle.locus_analyze(lf, rf, a1, [b"CAG", b"CCG"], reads[:1], genotyper=1)
    assert r2["n_alleles"] == 2 and r2["alleles"] == [a1.decode()] * 2 and r2["SD"] == "1,0"


def test_cluster_genotyper_size_and_cluster_agree_on_clean_locus(oracle):
    rng = np.random.default_rng(11)
    a1, a2 = b"GAA" * 12, b"GAA" * 30
    lf, rf, reads = _locus(rng, [a1, a2], 20)
    rs = oracle.locus_analyze(lf, rf, a1, [b"GAA"], reads, genotyper=0)
    rc = oracle.locus_analyze(lf, rf, a1, [b"GAA"], reads, genotyper=1)
    assert rs["alleles"] == rc["alleles"] and rs["MC"] == rc["MC"] == "12,30"


def test_filter_impure_trs(oracle):
    rng = np.random.default_rng(5)
    pure = b"CAG" * 20
    impure = bytearray(pure)
    for i in range(0, 60, 5):
        impure[i] = ord("T")
    impure = bytes(impure)
    lf, rf, reads = _locus(rng, [pure], 12)
    lf2, rf2 = lf, rf
    bad = reads[0][:500] + impure + reads[0][500 + len(pure):]
    reads[3] = bad
    reads[7] = bad
    # default min_read_qual 0.98: the filter is off
    r = oracle.locus_analyze(lf, rf, pure, [b"CAG"], reads)
    assert len(r["kept_read"]) == 12 and r["stats"]["n_purity"] == 0
    # min_read_qual < 0.9 and no rq tag on any read: every read goes through the HMM, at most max(1, round(1.2)) = 1 is dropped,
    # and the survivors come back ordered by purity (the second impure read first)
    r = oracle.locus_analyze(lf2, rf2, pure, [b"CAG"], reads, min_read_qual=0.5)
    assert r["stats"]["n_purity"] == 12
    assert len(r["kept_read"]) == 11 and r["kept_read"][0] == 7 and 3 not in r["kept_read"].tolist()
    # high-quality reads are not looked at
    rq = np.full(12, 0.999)
    rq[7] = 0.5
    r = oracle.locus_analyze(lf2, rf2, pure, [b"CAG"], reads, min_read_qual=0.5, read_qual=rq)
    assert r["stats"]["n_purity"] == 1 and 7 not in r["kept_read"].tolist() and len(r["kept_read"]) == 11
