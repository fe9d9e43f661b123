"""Parity of the GPU locus path (trgt_find_spans_batch / trgt_locus_batch through the C ABI) against the CPU oracle's
restatement of analyze_tr on seeded synthetic loci: spans per read, kept reads and their order,
allele strings, CI, classification, MC / MS / AP."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from trgt_amd import locus, synth
    return locus, synth


def _lib_mod():
    from trgt_amd import _lib
    return _lib


def _run_both(locus, b, params=None):
    """trgt_locus_batch three ways -- host glue for every locus (TRGT_HOST_GENOTYPER), reads on the host with the device genotyper
    working on the uploaded copy, reads resident in HBM -- all must give the oracle's results."""
    import torch
    from trgt_amd import _lib
    params = params or locus.Params()
    hctx = _lib.context_with_env(TRGT_HOST_GENOTYPER=1)  # (planner knobs are read when a context is created)
    try:
        out = locus.run_batch(b, params, ctx=hctx)
    finally:
        hctx.close()
    yield "host glue", out
    rctx = _lib.context_with_env(TRGT_HOST_REPAIR=1)  # loci whose pick lacks majority support go back to the host (no device-side repair)
    try:
        out = locus.run_batch(b, params, ctx=rctx)
        assert int(out.stats[18]) == 0
    finally:
        rctx.close()
    yield "host repair", out
    sctx = _lib.context_with_env(TRGT_REPAIR_MAX_SEG=60)  # device-side repair for short segments only: both paths in one call
    try:
        out = locus.run_batch(b, params, ctx=sctx)
    finally:
        sctx.close()
    yield "mixed repair", out
    pctx = _lib.context_with_env(TRGT_SPLIT_HMM=1)  # the HMM of the settled loci next to the device-side repair, a second batch behind it
    try:
        out = locus.run_batch(b, params, ctx=pctx)
    finally:
        pctx.close()
    yield "split hmm", out
    cctx = _lib.context_with_env(TRGT_HOST_CLUSTER=1)  # Genotyper::Cluster loci on host threads (locus_cluster.hpp), not by the device chain
    try:
        out = locus.run_batch(b, params, ctx=cctx)
        assert int(out.stats[22]) == 0
    finally:
        cctx.close()
    yield "host cluster", out
    dctx = _lib.context_with_env(TRGT_HMM_NO_DEDUPE=1, TRGT_HMM_NO_LONG_TB=1, TRGT_HMM_LDS_FILL=1)  # every allele labelled by a job of its own, the round-3 HMM paths
    try:
        out = locus.run_batch(b, params, ctx=dctx)
    finally:
        dctx.close()
    yield "plain hmm", out
    qctx = _lib.context_with_env(TRGT_HMM_PPL_PER_CLASS=1)  # the position-per-lane fills launched class by class (not once per width over all classes)
    try:
        out = locus.run_batch(b, params, ctx=qctx)
    finally:
        qctx.close()
    yield "fills per class", out
    yield "host reads", locus.run_batch(b, params)
    reads_dev = torch.from_numpy(b["read_blob"]).cuda()
    flank_dev = torch.from_numpy(b["flank_blob"]).cuda()
    yield "device", locus.run_batch(b, params, flank_dev=flank_dev, reads_dev=reads_dev)


def _oracle_locus(oracle, b, l, params):
    a0, a1 = int(b["locus_read_begin"][l]), int(b["locus_read_begin"][l + 1])
    reads = [bytes(b["read_blob"][int(b["read_off"][r]):int(b["read_off"][r]) + int(b["read_len"][r])]) for r in range(a0, a1)]
    lf = bytes(b["flank_blob"][int(b["lf_off"][l]):int(b["lf_off"][l]) + int(b["lf_len"][l])])
    rf = bytes(b["flank_blob"][int(b["rf_off"][l]):int(b["rf_off"][l]) + int(b["rf_len"][l])])
    tr = bytes(b["tr_blob"][int(b["tr_off"][l]):int(b["tr_off"][l]) + int(b["tr_len"][l])])
    m0, m1 = int(b["set_motif_begin"][l]), int(b["set_motif_begin"][l + 1])
    motifs = [bytes(b["motif_blob"][int(b["motif_off"][m]):int(b["motif_off"][m + 1])]) for m in range(m0, m1)]
    gt = int(b["genotyper"][l]) if b.get("genotyper") is not None else 0
    rq = b["read_qual"][a0:a1] if b.get("read_qual") is not None else None
    return oracle.locus_analyze(lf, rf, tr, motifs, reads, flank_len=params.search_flank_len,
                                min_flank_id_frac=params.min_flank_id_frac, max_depth=params.max_depth,
                                scoring=params.aln_scoring, ploidy=int(b["ploidy"][l]), genotyper=gt,
                                min_read_qual=params.min_read_qual, read_qual=rq)


def _compare(oracle, locus, b, out, params, loci):
    n_repair = 0
    for l in loci:
        ref = _oracle_locus(oracle, b, l, params)
        a0, a1 = int(b["locus_read_begin"][l]), int(b["locus_read_begin"][l + 1])
        assert np.array_equal(out.span_start[a0:a1], ref["span_start"]), l
        assert np.array_equal(out.span_end[a0:a1], ref["span_end"]), l
        got = locus.locus_result(b, out, l)
        assert len(got.genotype) == ref["n_alleles"], l
        assert [a.seq.decode() for a in got.genotype] == ref["alleles"], l
        assert got.reads == [int(v) for v in ref["kept_read"]], l
        assert got.classification == [int(v) for v in ref["classification"]], l
        if ref["n_alleles"]:
            f = got.vcf_fields()
            for k in ("AL", "ALLR", "SD", "MC", "MS", "AP"):
                assert f[k] == ref[k], (l, k)
        n_repair += ref["stats"]["n_wfa_cons"] > 0
    return n_repair


def test_find_spans_matches_oracle(oracle, mods):
    locus, synth = mods
    b = synth.generate(24, first_locus=1000)
    ss, se, lh, rh = locus.find_tr_spans_batch(b)
    for l in range(24):
        ref = _oracle_locus(oracle, b, l, locus.Params())
        a0, a1 = int(b["locus_read_begin"][l]), int(b["locus_read_begin"][l + 1])
        assert np.array_equal(ss[a0:a1], ref["span_start"]) and np.array_equal(se[a0:a1], ref["span_end"])
    # every complete read of a clean synthetic locus spans; reads cut inside a flank or the repeat cannot
    assert (ss[b["read_truncated"] == 0] >= 0).mean() > 0.95
    assert (ss[b["read_truncated"] == 1] >= 0).mean() < 0.75
    assert set(np.unique(lh)) <= {0, 1, 2}


def test_locus_batch_matches_oracle_cfg2(oracle, mods):
    locus, synth = mods
    b = synth.generate(160, first_locus=0)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(160))
        # genotype sanity against the generator's ground truth: allele lengths recovered for nearly every locus
        ok = 0
        for l in range(160):
            got = sorted(int(v) for v in out.allele_len[2 * l:2 * l + 2])
            ok += got == sorted(int(v) for v in b["true_allele_len"][2 * l:2 * l + 2])
        assert ok >= 120, mode


def test_locus_batch_noisy_reads_trigger_consensus_repair(oracle, mods):
    # high error + stutter rates: no majority sequence -> utils::align (BiWFA) + repair_consensus path
    locus, synth = mods
    b = synth.generate(60, first_locus=5000, sub_rate=0.004, ins_rate=0.004, del_rate=0.004, stutter_rate=0.3)
    for mode, out in _run_both(locus, b):
        n_repair = _compare(oracle, locus, b, out, locus.Params(), range(60))
        assert n_repair > 5 and int(out.stats[1]) > 0, mode
        if mode in ("host reads", "device"):  # the consensus repair ran on the device (locus_gt.hpp), nothing found no room
            assert int(out.stats[18]) > 5 and int(out.stats[19]) == 0 and int(out.stats[20]) == int(out.stats[1]), (mode, out.stats[18:21])


def test_locus_batch_device_resident_reads_and_downsampling(oracle, mods):
    import torch
    locus, synth = mods
    b = synth.generate(12, first_locus=777, reads_per_locus=80)
    params = locus.Params(max_depth=25)
    reads_dev = torch.from_numpy(b["read_blob"]).cuda()
    flank_dev = torch.from_numpy(b["flank_blob"]).cuda()
    out = locus.run_batch(b, params, flank_dev=flank_dev, reads_dev=reads_dev)
    _compare(oracle, locus, b, out, params, range(12))
    assert all(int((out.read_rank[int(b["locus_read_begin"][l]):int(b["locus_read_begin"][l + 1])] >= 0).sum()) <= 25 for l in range(12))


def test_empty_and_degenerate_loci(oracle, mods):
    locus, _ = mods
    rng = np.random.default_rng(1)
    dna = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    lf, rf = dna(250), dna(250)
    loci = [dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 5, motifs=[b"CAG"], reads=[]),                    # no reads
            dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 5, motifs=[b"CAG"], reads=[dna(300), dna(40)]),   # nothing spans
            dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 5, motifs=[b"CAG"], ploidy=1,
                 reads=[dna(260) + lf + b"CAG" * k + rf + dna(255) for k in (5, 5, 6, 5)]),                   # haploid
            dict(left_flank=lf, right_flank=rf, tr=b"", motifs=[b"A"], reads=[dna(250) + lf + rf + dna(250)] * 3)]  # empty repeat
    res = locus.analyze_batch(loci)
    assert res[0].genotype == [] and res[1].genotype == []
    assert len(res[2].genotype) == 1 and res[2].genotype[0].seq == b"CAG" * 5 and res[2].vcf_fields()["MC"] == "5"
    assert [a.seq for a in res[3].genotype] == [b"", b""] and res[3].vcf_fields()["AP"] == ".,." and res[3].vcf_fields()["MS"] == ".,."
    b = locus.pack(loci)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(4))


def test_reference_example_locus_matches_tutorial_vcf(oracle, mods):
    # the reference's own end-to-end golden (docs/tutorial.md:29-46): reads of example/sample.bam, clipped as analyze_tr does
    from test_oracle_example_e1 import load_e1
    locus, _ = mods
    e1, want = load_e1()
    L = dict(left_flank=e1["left_flank"].encode(), right_flank=e1["right_flank"].encode(), tr=e1["tr"].encode(),
             motifs=[m.encode() for m in e1["motifs"]], reads=[r.encode() for r in e1["reads"]])
    res = locus.analyze_batch([L, L])
    for r in res:
        assert [a.seq.decode() for a in r.genotype] == [want["alt"], want["alt"]]
        f = r.vcf_fields()
        for k in ("AL", "ALLR", "SD", "MC", "MS", "AP"):
            assert f[k] == want[k], k
    b = locus.pack([L])
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(1))
        assert {k: v for k, v in locus.locus_result(b, out, 0).vcf_fields().items() if k in want} == {k: want[k] for k in ("AL", "ALLR", "SD", "MC", "MS", "AP")}, mode


def test_flank_scan_shapes(oracle, mods):
    # the scan kernel walks a read in rounds of 1024 bytes, 16 candidate windows per lane: reads around the round size, flanks at
    # the very ends, repeated flank heads (many false 4-byte candidates) and flank lengths that are not a multiple of 4
    locus, _ = mods
    rng = np.random.default_rng(7)
    dna = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    for F in (250, 37, 6):
        lf, rf = dna(F), dna(F)
        lf = lf[:4] * 3 + lf[12:] if F > 16 else lf          # the head of the left piece repeats inside it
        reads = []
        for n_left, n_tr, n_right in ((0, 30, 0), (1, 0, 1), (700, 300, 10), (1024 - F, 5, 1024), (1500, 60, 1500), (2, 9, 3000), (1023, 1, 1)):
            reads.append(dna(n_left) + lf + b"CAG" * (n_tr // 3) + rf + dna(n_right))
        reads.append(lf[:4] * 40 + lf + b"CAG" * 4 + rf)      # decoy heads in front of the true occurrence
        reads.append(dna(300) + lf[:-1] + b"N" + b"CAG" * 4 + rf)  # left piece present with one mismatch only
        reads.append(lf[: F - 1])                               # shorter than the flank
        loci = [dict(left_flank=dna(20) + lf, right_flank=rf + dna(20), tr=b"CAG" * 5, motifs=[b"CAG"], reads=reads)]
        b = locus.pack(loci)
        params = locus.Params(search_flank_len=F)
        ss, se, lh, rh = locus.find_tr_spans_batch(b, params)
        ref = oracle.locus_analyze(loci[0]["left_flank"], loci[0]["right_flank"], loci[0]["tr"], loci[0]["motifs"], reads, flank_len=F,
                                   min_flank_id_frac=params.min_flank_id_frac, max_depth=params.max_depth, scoring=params.aln_scoring)
        assert np.array_equal(ss, ref["span_start"]) and np.array_equal(se, ref["span_end"]), F


def test_device_genotyper_envelope(oracle, mods):
    # loci the device genotyper must hand back or treat specially: haploid, no spanning read, > 256 reads, repeat segments that do
    # not fit its LDS staging (long alleles), homozygous and well separated heterozygous loci, ties between allele lengths
    locus, _ = mods
    rng = np.random.default_rng(11)
    dna = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    lf, rf = dna(250), dna(250)
    mk = lambda k, m=b"CAG": dna(int(rng.integers(250, 300))) + lf + m * k + rf + dna(int(rng.integers(250, 300)))
    loci = [
        dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 5, motifs=[b"CAG"], ploidy=1, reads=[mk(k) for k in (7, 7, 8, 7, 6, 7)]),
        dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 5, motifs=[b"CAG"], reads=[dna(700) for _ in range(5)]),
        dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 9, motifs=[b"CAG"], reads=[mk(9 if i % 2 else 14) for i in range(300)]),
        dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 5, motifs=[b"CAG"], reads=[mk(700 + (i % 2)) for i in range(12)]),
        dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 10, motifs=[b"CAG"], reads=[mk(10) for _ in range(20)]),
        dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 10, motifs=[b"CAG"], reads=[mk(10 if i < 10 else 40) for i in range(20)]),
        dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 11, motifs=[b"CAG"], reads=[mk(k) for k in (10, 12, 10, 12, 11, 10, 12)]),
        dict(left_flank=lf, right_flank=rf, tr=b"AT" * 6, motifs=[b"AT"], reads=[mk(k, b"AT") for k in (5, 6, 7, 5, 6, 7)]),
    ]
    b = locus.pack(loci)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(len(loci)))


def test_cluster_genotyper_cfg5_matches_oracle(oracle, mods):
    # compound / N-containing motif sets, Genotyper::Cluster -- n(n-1)/2 edit-distance alignments,
    # Ward linkage, a consensus alignment for every read, outlier assignment (genotype_cluster.rs:58-152)
    locus, synth = mods
    b = synth.generate(48, first_locus=300, config=5)
    assert b["genotyper"].all()
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(48))
        assert int(out.stats[15]) > 0 and int(out.stats[1]) > 48 * 10, mode  # edit-distance and consensus jobs ran on the GPU
        if mode in ("host reads", "device", "host repair", "mixed repair"):  # ... and the linkage, the groups and the rounds on the device as well
            assert int(out.stats[22]) == 48 and int(out.stats[23]) == 0, (mode, out.stats[22:24])
    # arenas too small for all loci: some are genotyped by the device chain, the others find no room and take the host path in the same call
    actx = _lib_mod().context_with_env(TRGT_CLUSTER_ARENA_KB=600)
    try:
        out = locus.run_batch(b, locus.Params(), ctx=actx)
    finally:
        actx.close()
    _compare(oracle, locus, b, out, locus.Params(), range(48))
    assert 0 < int(out.stats[22]) < 48 and int(out.stats[23]) > 0, out.stats[22:24]
    # short alleles: |a|*|b| <= MAX_OPS for every pair, so the whole distance matrix comes from edit-distance alignments
    b = synth.generate(24, first_locus=7000, config=5, max_allele_bp=90)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(24))
        assert int(out.stats[15]) > 24 * 200, mode
    # noisy reads: outlier clusters, consensus repair with insertions
    b = synth.generate(24, first_locus=900, config=5, sub_rate=0.004, ins_rate=0.004, del_rate=0.004, stutter_rate=0.3)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(24))


def test_cluster_genotyper_shapes(oracle, mods):
    # the branches of genotype_cluster::genotype: haploid, a single read, two reads, homozygous (even / odd split), a small far
    # group of outlier reads, alleles long enough for the |a|*|b| > MAX_OPS shortcut, size and cluster loci mixed in one batch
    locus, _ = mods
    rng = np.random.default_rng(23)
    dna = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    lf, rf = dna(250), dna(250)
    mk = lambda rep: dna(int(rng.integers(250, 300))) + lf + rep + rf + dna(int(rng.integers(250, 300)))
    noisy = lambda rep: bytes(int(rng.choice(list(b"ACGT"))) if rng.random() < 0.03 else c for c in rep)
    base = dict(left_flank=lf, right_flank=rf, motifs=[b"CAG", b"CCG"], genotyper="cluster")
    loci = [
        dict(base, tr=b"CAG" * 8, ploidy=1, reads=[mk(noisy(b"CAG" * 8)) for _ in range(9)]),
        dict(base, tr=b"CAG" * 8, reads=[mk(b"CAG" * 9)]),
        dict(base, tr=b"CAG" * 8, reads=[mk(b"CAG" * 9), mk(b"CAG" * 12)]),
        dict(base, tr=b"CAG" * 8, reads=[mk(b"CAG" * 8) for _ in range(11)]),
        dict(base, tr=b"CAG" * 8, reads=[mk(noisy(b"CAG" * 8 + b"CCG" * (3 if i % 2 else 9))) for i in range(24)]),
        dict(base, tr=b"CAG" * 8, reads=[mk(b"CAG" * (8 if i < 14 else 11)) for i in range(16)] + [mk(noisy(b"CAG" * 30)), mk(dna(20))]),
        dict(base, tr=b"CAG" * 60, reads=[mk(noisy(b"CAG" * (60 if i % 2 else 75))) for i in range(14)]),
        dict(base, tr=b"CAG" * 8, genotyper="size", reads=[mk(b"CAG" * (8 if i % 2 else 10)) for i in range(12)]),
        dict(base, tr=b"CAG" * 10, reads=[mk(b"CAG" * 10) for _ in range(13)] + [mk(b"CAG" * 12)]),
        dict(base, tr=b"", motifs=[b"A"], reads=[mk(b"") for _ in range(4)]),
    ]
    b = locus.pack(loci)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(len(loci)))
        if mode in ("host reads", "device"):
            assert int(out.stats[22]) == len(loci) - 1 and int(out.stats[23]) == 0, (mode, out.stats[22:24])
    res = locus.analyze_batch(loci)
    assert len(res[0].genotype) == 1 and len(res[1].genotype) == 2 and res[1].genotype[0].seq == res[1].genotype[1].seq == b"CAG" * 9
    assert [len(a.seq) for a in res[2].genotype] == [27, 36]


def test_cluster_empty_segment_against_segments_beyond_10kb(oracle, mods):
    # |a| * |b| <= MAX_OPS holds for an EMPTY repeat segment against one of any length, so the edit-distance launch of the
    # device chain sees (0, > 10 kb) pairs: its workspace must be planned for them (the rings were sized for 10 kb + 1)
    locus, _ = mods
    rng = np.random.default_rng(5)
    dna = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    lf, rf = dna(250), dna(250)
    mk = lambda rep: dna(260) + lf + rep + rf + dna(260)
    noisy = lambda rep: bytes(int(rng.choice(list(b"ACGT"))) if rng.random() < 0.002 else c for c in rep)
    long_rep = b"CAG" * 3600   # 10 800 bases
    base = dict(left_flank=lf, right_flank=rf, motifs=[b"CAG"], genotyper="cluster", tr=b"CAG" * 8)
    loci = [
        dict(base, reads=[mk(b"") for _ in range(4)] + [mk(noisy(long_rep)) for _ in range(4)]),
        dict(base, reads=[mk(b"") for _ in range(3)] + [mk(noisy(long_rep[:10200])) for _ in range(3)] + [mk(b"CAG" * 9) for _ in range(3)]),
    ]
    b = locus.pack(loci)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(len(loci)))
        if mode in ("host reads", "device"):
            assert int(out.stats[22]) == len(loci), (mode, out.stats[22:24])


def test_cluster_genotyper_deep_loci_and_downsampling(oracle, mods):
    # more than 64 reads per locus: the large instantiations of the device chain (distance matrix in HBM); more reads than max_depth:
    # the uniform downsample in front of the pair list; a locus beyond 256 reads takes the host path
    locus, synth = mods
    b = synth.generate(6, first_locus=40, config=5, reads_per_locus=100, max_allele_bp=120)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(6))
        if mode in ("host reads", "device"):
            assert int(out.stats[22]) == 6, (mode, out.stats[22:24])
    params = locus.Params(max_depth=40)
    for mode, out in _run_both(locus, b, params):
        _compare(oracle, locus, b, out, params, range(6))
    b = synth.generate(2, first_locus=90, config=5, reads_per_locus=280, max_allele_bp=60)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(2))
        assert int(out.stats[22]) == 0, mode


def test_filter_impure_trs_matches_oracle(oracle, mods):
    # --min-read-quality below 0.9 switches the HMM purity filter on (tr.rs:37-50, 400-452): reads without an rq >= 0.9 are scored,
    # the list is re-ordered by purity and at most max(1, round(0.1 n)) impure reads are dropped
    locus, synth = mods
    b = synth.generate(40, first_locus=4000, sub_rate=0.01, ins_rate=0.004, del_rate=0.004)
    rng = np.random.default_rng(3)
    nr = int(b["n_reads"])
    params = locus.Params(min_read_qual=0.5)
    for rq in (None, np.where(rng.random(nr) < 0.5, 0.999, np.where(rng.random(nr) < 0.5, 0.7, np.nan))):
        bb = dict(b)
        bb["read_qual"] = rq
        for mode, out in _run_both(locus, bb, params):
            _compare(oracle, locus, bb, out, params, range(40))
    # hand-made loci with impure repeats: some reads must be dropped, the rest re-ordered by purity
    dna = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    lf, rf = dna(250), dna(250)
    def impure(rep, k):
        rep = bytearray(rep)
        for i in rng.choice(len(rep), size=k, replace=False):
            rep[i] = ord("T") if rep[i] != ord("T") else ord("A")
        return bytes(rep)
    mk = lambda rep: dna(int(rng.integers(250, 300))) + lf + rep + rf + dna(int(rng.integers(250, 300)))
    loci = []
    for n_reads, n_bad in ((12, 2), (30, 5), (4, 4), (25, 1)):
        reads = [mk(impure(b"CAG" * 20, 14) if i < n_bad else impure(b"CAG" * 20, int(rng.integers(0, 3)))) for i in range(n_reads)]
        order = rng.permutation(n_reads)
        loci.append(dict(left_flank=lf, right_flank=rf, tr=b"CAG" * 20, motifs=[b"CAG"], reads=[reads[i] for i in order],
                         read_qual=[None if rng.random() < 0.5 else 0.6 for _ in range(n_reads)]))
    bh = locus.pack(loci)
    for mode, out in _run_both(locus, bh, params):
        _compare(oracle, locus, bh, out, params, range(len(loci)))
        assert int(((out.span_start >= 0) & (out.read_rank < 0)).sum()) >= 4, mode
    # cluster genotyper behind the filter
    b5 = synth.generate(10, first_locus=40, config=5, sub_rate=0.01)
    for mode, out in _run_both(locus, b5, params):
        _compare(oracle, locus, b5, out, params, range(10))


def _cfg3_loci(rng, sets, long_lo, long_hi, n_reads):
    """allele 1 = 10-40 copies in total, allele 2 = log-uniform total length built as consecutive runs
    of each motif (N filled uniformly), 1 % of the units carry a substitution; reads with HiFi-like errors."""
    from helpers import mutate, rand_dna
    fill = lambda m: bytes(b if b != ord("N") else int(rng.choice(list(b"ACGT"))) for b in m)
    def allele(motifs, total_len):
        out = bytearray()
        per = max(1, total_len // len(motifs))
        for m in motifs:
            run = bytearray()
            while len(run) < per:
                u = bytearray(fill(m))
                if rng.random() < 0.01:
                    u[int(rng.integers(0, len(u)))] = int(rng.choice(list(b"ACGT")))
                run += u
            out += run
        return bytes(out)
    loci = []
    for s in sets:
        motifs = [m.encode() for m in s["motifs"]]
        mean_len = sum(len(m) for m in motifs) / len(motifs)
        a1 = allele(motifs, int(int(rng.integers(10, 41)) * mean_len))
        a2 = allele(motifs, int(np.exp(rng.uniform(np.log(long_lo), np.log(long_hi)))))
        lf, rf = rand_dna(rng, 250), rand_dna(rng, 250)
        lc, rc = rand_dna(rng, 250), rand_dna(rng, 250)
        reads = [mutate(rng, lc + lf + (a1 if i % 2 else a2) + rf + rc, 5e-4, 2.5e-4, 2.5e-4) for i in range(n_reads)]
        reads.append(reads[0][: 250 + 250 + len(a2) // 2])  # a read that ends inside the repeat
        loci.append(dict(left_flank=lf, right_flank=rf, tr=a1, motifs=motifs, reads=reads))
    return loci


def test_cfg3_pathogenic_motif_sets(oracle, mods):
    # BASELINE configs[2]: the 56 motif sets of the reference's pathogenic catalog (tests/golden/pathogenic_motif_sets.json, from
    # repeats/pathogenic_repeats.hg38.bed) with one expanded allele each: multi-motif HMMs up to 170 states, alleles past the
    # LDS-staged sizes of the HMM and genotyper kernels, reads longer than the dedicated WFA kernel takes
    import json, os
    locus, _ = mods
    sets = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pathogenic_motif_sets.json")))["loci"]
    assert len(sets) == 56
    rng = np.random.default_rng(56)
    loci = _cfg3_loci(rng, sets, 500, 3000, 8)
    b = locus.pack(loci)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(len(loci)))
    res = locus.analyze_batch(loci[:4])
    assert all(len(r.genotype) == 2 for r in res)


def test_cfg3_alleles_to_10kb(oracle, mods):
    # the upper end of cfg3: 10 kb alleles, RFC1's ten-motif set (170 HMM states) and two single-motif loci
    import json, os
    locus, _ = mods
    sets = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pathogenic_motif_sets.json")))["loci"]
    pick = [s for s in sets if s["id"] in ("RFC1", "HTT", "CNBP")]
    rng = np.random.default_rng(10)
    loci = _cfg3_loci(rng, pick, 9000, 10000, 6)
    b = locus.pack(loci)
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(len(loci)))
        assert int(out.allele_len.max()) > 8500, mode


def test_flank_launch_variants_agree(oracle, mods, monkeypatch):
    # the planner's choices must not show in the results: one launch instead of two for the flank alignments, the general
    # instantiation of the dedicated kernel instead of the compile-time flank configuration, the host genotyper, other numbers of
    # waves per alignment, no seeded windows / other numbers of segments for them, no band / other bands for the back-trace of what the
    # pre-filter keeps
    import torch
    locus, synth = mods
    b = synth.generate(96, first_locus=12000)
    rd, fd = torch.from_numpy(b["read_blob"]).cuda(), torch.from_numpy(b["flank_blob"]).cuda()
    base = locus.run_batch(b, flank_dev=fd, reads_dev=rd)
    _compare(oracle, locus, b, base, locus.Params(), range(0, 96, 4))
    for env, val in (("TRGT_WFA_ONE_LAUNCH", "1"), ("TRGT_WFA_NO_SPEC", "1"), ("TRGT_HOST_GENOTYPER", "1"), ("TRGT_HEAVY_THREADS", "256"),
                     ("TRGT_HEAVY_THREADS", "128"), ("TRGT_FLANK_THREADS", "192"), ("TRGT_WFA_NO_WINDOW", "1"), ("TRGT_WIN_SEGMENTS", "4"),
                     ("TRGT_WIN_SEGMENTS", "6"), ("TRGT_WIN_THREADS", "128"), ("TRGT_WFA_NO_FILTER", "1"), ("TRGT_FILTER_PER_CU", "3"), ("TRGT_HEAVY_BAND", "0"), ("TRGT_HEAVY_BAND", "20"),
                     ("TRGT_HEAVY_BAND", "256"),
                     # round 5: other widths of the banded back-trace's workgroups, the per-launch hipMemsetAsync path in place of the
                     # zero arena, the round-4 HMM fills, the position-per-lane fills of a class one after the other, other claim sizes and
                     # the 128-diagonal tier of the lean kernel
                     ("TRGT_BAND_THREADS", "128"), ("TRGT_BAND_THREADS", "256"), ("TRGT_NO_ZERO_ARENA", "1"), ("TRGT_HMM_NO_PPL", "1"),
                     ("TRGT_HMM_PPL_SERIAL", "1"), ("TRGT_HMM_PPL_PER_CLASS", "1"), ("TRGT_HMM_PPL_WIDE", "1"), ("TRGT_LEAN_CHUNK", "1"), ("TRGT_LEAN_CHUNK", "16"), ("TRGT_WFA_LEAN_MID_TIER", "1")):
        from trgt_amd import _lib
        vctx = _lib.context_with_env(**{env: val})
        try:
            out = locus.run_batch(b, flank_dev=fd, reads_dev=rd, ctx=vctx)
        finally:
            vctx.close()
        env = env + "=" + val
        for f in ("span_start", "span_end", "n_alleles", "allele_len", "ci", "num_spanning", "classification", "read_rank", "n_spans"):
            assert np.array_equal(getattr(out, f), getattr(base, f)), (env, f)
        assert np.array_equal(out.purity.view(np.uint64), base.purity.view(np.uint64)), env
        for l in range(96):
            assert locus.locus_result(b, out, l).vcf_fields() == locus.locus_result(b, base, l).vcf_fields(), (env, l)


def test_cfg4_catalog_mix_matches_oracle(oracle, mods):
    # BASELINE configs[3] stand-in: 70 % single STR loci, 20 % loci with 2-5 motifs, 10 % VNTR loci with one
    # motif of 7-60 bp and alleles up to 600 bp -- HMMs from 10 to ~190 states in one batch, size genotyper
    locus, synth = mods
    b = synth.generate(120, first_locus=2024, config=4)
    nm = np.diff(b["set_motif_begin"])
    mlen = np.diff(b["motif_off"])
    assert (nm > 1).any() and (mlen >= 20).any() and (nm == 1).sum() > 60
    for mode, out in _run_both(locus, b):
        _compare(oracle, locus, b, out, locus.Params(), range(120))
