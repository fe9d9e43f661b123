"""Host-side logic that needs no GPU: synthetic generator (determinism, shard locality), batch packing, the
WFAligner / HMM mirrors' pure-python pieces, and the oracle's locus pipeline on generated data."""
import numpy as np
import pytest


def test_synth_is_deterministic_and_shard_local():
    from trgt_amd import synth
    a = synth.generate(12, first_locus=0)
    b = synth.generate(12, first_locus=0, threads=3)
    for k in ("read_blob", "flank_blob", "motif_blob", "read_len", "true_allele_len"):
        assert np.array_equal(a[k], b[k]), k
    s = synth.generate(5, first_locus=4)
    r0, r1 = int(a["locus_read_begin"][4]), int(a["locus_read_begin"][9])
    lo, hi = int(a["read_off"][r0]), int(a["read_off"][r1 - 1]) + int(a["read_len"][r1 - 1])
    assert np.array_equal(s["read_blob"], a["read_blob"][lo:hi])
    assert np.array_equal(s["true_allele_len"], a["true_allele_len"][8:18])
    assert np.array_equal(s["motif_blob"], a["motif_blob"][int(a["motif_off"][4]):int(a["motif_off"][9])])


def test_synth_matches_appendix_e_shape():
    from trgt_amd import synth
    b = synth.generate(200)
    assert b["n_reads"] == 200 * 30 and (b["lf_len"] == 250).all() and (b["rf_len"] == 250).all()
    mlen = np.diff(b["motif_off"])
    assert mlen.min() >= 3 and mlen.max() <= 6
    assert b["true_allele_len"].max() <= 200 and b["true_allele_len"].min() >= 9
    assert 0.05 < b["read_truncated"].mean() < 0.16
    full = b["read_len"][b["read_truncated"] == 0]
    assert full.min() > 1000 and full.max() < 1260
    assert set(np.unique(b["read_blob"])) <= set(b"ACGT")
    hap = b["read_hap"].reshape(200, 30)
    assert (hap.sum(1) >= 5).all() and (hap.sum(1) <= 25).all()


def test_oracle_genotypes_synthetic_loci(oracle):
    """The oracle's analyze_tr restatement recovers the generator's truth (sanity of both)."""
    from trgt_amd import synth
    b = synth.generate(25, first_locus=300)
    ok = 0
    for l in range(25):
        a0, a1 = int(b["locus_read_begin"][l]), int(b["locus_read_begin"][l + 1])
        reads = [bytes(b["read_blob"][int(b["read_off"][r]):int(b["read_off"][r]) + int(b["read_len"][r])]) for r in range(a0, a1)]
        lf = bytes(b["flank_blob"][int(b["lf_off"][l]):int(b["lf_off"][l]) + 250])
        rf = bytes(b["flank_blob"][int(b["rf_off"][l]):int(b["rf_off"][l]) + 250])
        tr = bytes(b["tr_blob"][int(b["tr_off"][l]):int(b["tr_off"][l]) + int(b["tr_len"][l])])
        motifs = [bytes(b["motif_blob"][int(b["motif_off"][l]):int(b["motif_off"][l + 1])])]
        r = oracle.locus_analyze(lf, rf, tr, motifs, reads)
        assert r["n_alleles"] == 2 and len(r["kept_read"]) >= 20
        assert (r["span_start"][b["read_truncated"][a0:a1] == 0] >= 0).all()
        ok += sorted(len(x) for x in r["alleles"]) == sorted(int(v) for v in b["true_allele_len"][2 * l:2 * l + 2])
        for ap in r["AP"].split(","):
            assert float(ap) > 0.85
    assert ok >= 20


def test_pack_layout_and_outputs_shape():
    from trgt_amd import locus
    loci = [dict(left_flank=b"A" * 250, right_flank=b"C" * 250, tr=b"CAGCAG", motifs=["CAG", "CCG"], reads=[b"ACGT" * 10, b"TTTT"]),
            dict(left_flank=b"G" * 250, right_flank=b"T" * 250, tr=b"", motifs=[b"A"], ploidy=1, reads=[])]
    b = locus.pack(loci)
    assert b["n_loci"] == 2 and b["n_reads"] == 2
    assert list(b["locus_read_begin"]) == [0, 2, 2] and list(b["set_motif_begin"]) == [0, 2, 3] and list(b["ploidy"]) == [2, 1]
    assert bytes(b["read_blob"][int(b["read_off"][1]):int(b["read_off"][1]) + 4]) == b"TTTT"
    o = locus.BatchOutputs(b)
    assert o.allele_cap[0] == 48 and o.allele_cap[1] == 8
    assert list(o.count_off) == [0, 2, 4, 5] and o.span_off[1] == 49


def test_wfaligner_mirror_pure_python_parts():
    from trgt_amd import wfaligner as W
    with pytest.raises(RuntimeError, match="penalty model"):
        W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryHigh).build()      # wfaligner.rs:361-363
    al = W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryLow).affine(12, 24, 2).with_heuristic(
        W.Heuristic.wfadaptive(10, 50, 100)).build()
    assert al.get_heuristics() == W.Heuristic.wfadaptive(10, 50, 100)                            # :1678-1716
    assert al.get_penalties() == W.Penalties("Affine", match_=0, mismatch=12, gap_opening=24, gap_extension=2)
    al.set_heuristic(W.Heuristic.banded_static(5, 20))
    assert al.get_heuristics() == W.Heuristic.banded_static(5, 20)
    assert W.WFAligner.decode_sam_cigar([183]) == [(11, "=")]                                    # :1609-1630
    assert W.WFAligner.decode_sam_cigar([135, 24, 39]) == [(8, "="), (1, "X"), (2, "=")]
    assert W.WFAligner.decode_sam_cigar([176]) == [(11, "M")]
    p = al._params("endsfree", 0, 0, -1, -1)
    assert (p.span, p.text_begin_free, p.text_end_free, p.memory_mode, p.heuristic) == (1, -1, -1, 2, 5)    # BandedStatic: its own number, the library answers TRGT_ERR_UNSUPPORTED
    sc = W.WFAligner.builder(W.AlignmentScope.Score, W.MemoryModel.MemoryUltraLow).edit().build()
    assert sc._params("end2end").metric == 1 and sc._params("end2end").heuristic == 1  # default wfadaptive stays (:374-376)
    sc._last = dict(ops=b"", n_match=0, cigar=[], span=[0, 0, 0, 0], score=3, plen=0, tlen=0)
    with pytest.raises(RuntimeError):
        sc.count_matches()                                                                       # :989-991
    assert sc.cigar_operations() == b""                                                          # :911-913


def test_hmm_mirror_pure_python_parts():
    from trgt_amd import hmm as H
    assert H.replace_invalid_bases("ACGTNRY", "ATCG") == b"ACGTATC"                              # utils.rs:29-42
    assert H.replace_invalid_bases("GCN", "ATCGN") == b"GCN"
    assert H.num_states(["CAG"]) == 17 and H.num_states(["CAG", "CCG"]) == 27
    sp = [H.Span(0, 0, 3), H.Span(0, 3, 6), H.Span(1, 6, 9), H.Span(0, 10, 13)]
    assert H.collapse_labels(sp) == [H.Span(0, 0, 6), H.Span(1, 6, 9), H.Span(0, 10, 13)]        # utils.rs:11-27
    assert H.count_motifs(2, sp) == [3, 1]
    anns = [H.Annotation([H.Span(0, 0, 33)], [11], 1.0), H.Annotation(None, [0], float("nan"))]
    assert (H.encode_mc(anns), H.encode_ms(anns), H.encode_ap(anns)) == ("11,0", "0(0-33),.", "1.000000,.")
    b = H.pack_hmm_batch([["CAG"], ["A", "GCN"]], [(0, "CAGCAG"), (1, ""), (1, "GCAGCC")])
    assert list(b["seq_len"]) == [6, 0, 6] and list(b["count_off"]) == [0, 1, 3, 5] and list(b["span_off"]) == [0, 7, 8, 15]


def test_bench_host_plan_for_eight_ranks_stays_within_the_cpu_quota():
    """what bench.py plans for the first real 8-GPU run (simulated: WORLD_SIZE = 8 on the pool's hosts -- 256 CPUs, a cgroup
    quota of 16 -- and on an unrestricted host): ranks x contexts x host threads never exceeds what the process group may use (one thread
    per context is the floor), and the library's GPU waits only spin where the ranks' spinning fits the quota."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_plan", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for world in (1, 2, 4, 8):
        for contexts in (1, 4, 6):
            for cores, quota in ((256, 16.0), (256, None), (8, None), (64, 48.0)):
                threads, waits = bench.plan_host(world, contexts, cores, quota)
                allowed = min(cores, int(quota)) if quota else cores
                assert 1 <= threads <= 8
                assert threads * contexts * world <= max(allowed, contexts * world), (world, contexts, cores, quota, threads)
                spinning_cpus = world * 5.5
                if quota and spinning_cpus > quota:
                    assert waits.get("TRGT_POLL_WAIT") == "1", (world, quota)
                    assert waits["TRGT_POLL_SPIN_US"] in ("0", "200") and (waits["TRGT_POLL_SPIN_US"] == "0") == (world * 2.3 > quota)
                else:
                    assert waits == {}
    # the pool's hosts at N = 8: polling without spinning, one host thread per context
    threads, waits = bench.plan_host(8, 4, 256, 16.0)
    assert threads == 1 and waits == {"TRGT_POLL_WAIT": "1", "TRGT_POLL_SPIN_US": "0", "TRGT_POLL_NAP_US": "100"}
    # an explicit --host-threads is honoured up to the library's cap
    assert bench.plan_host(1, 4, 256, 16.0, 3)[0] == 3 and bench.plan_host(1, 4, 256, 16.0, 64)[0] == 8
