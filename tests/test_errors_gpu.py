"""Error behaviour at the C ABI: every entry point returns a negative TRGT_ERR_* with a message in trgt_hip_last_error, never aborts,
and leaves the context usable (include/trgt_hip.h "Errors"; the reference returns i32 statuses / Result<T, String> at the same places:
wfaligner.rs:148-159, utils/util.rs:3)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INVALID, UNSUPPORTED, NOMEM = -1, -3, -5


@pytest.fixture()
def env():
    from trgt_amd import _lib, hmm, locus, synth, wfaligner
    return _lib, hmm, locus, synth, wfaligner, _lib.Context(0)


def _wfa_call(_lib, ctx, p, pats, txts, seqs_null=False):
    n = len(pats)
    blob = np.frombuffer(b"".join(pats) + b"".join(txts) + b"A", np.uint8).copy()
    plen = np.array([len(x) for x in pats], np.uint32)
    tlen = np.array([len(x) for x in txts], np.uint32)
    po = np.zeros(n, np.uint64); po[1:] = np.cumsum(plen[:-1], dtype=np.uint64)
    to = np.zeros(n, np.uint64); to[1:] = np.cumsum(tlen[:-1], dtype=np.uint64); to += np.uint64(int(plen.sum()))
    status, score, nm = np.full(n, 99, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    span4 = np.zeros(4 * n, np.uint32)
    q = _lib.ptr
    rc = _lib.lib().trgt_wfa_batch(ctx.handle, C.byref(p), n, None if seqs_null else q(blob), q(po), q(plen), q(to), q(tlen), q(status), q(score),
                                   q(nm), q(span4), None, None, None, None, None, None)
    return rc, status, score


def _msg(_lib, ctx):
    return _lib.lib().trgt_hip_last_error(ctx.handle).decode()


def test_wfa_batch_rejects_bad_requests_and_recovers(env):
    _lib, hmm, locus, synth, W, ctx = env

    def params(**kw):
        p = _lib.WfaParams()
        _lib.lib().trgt_wfa_default_params(C.byref(p))
        p.metric, p.mismatch, p.gap_open1, p.gap_ext1 = 3, 2, 5, 1
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    pats, txts = [b"ACGTACGTAC", b"GGGG"], [b"ACGTTCGTAC", b"GGCGG"]
    # (heuristics 2 .. 6 = WFmash, XDrop, ZDrop, BandedStatic, BandedAdaptive: the header's contract for the rest of the enum, wfaligner.rs:707-780)
    for kw, want, word in ((dict(metric=7), INVALID, "metric"), (dict(heuristic=2), UNSUPPORTED, "Heuristic"), (dict(heuristic=3), UNSUPPORTED, "Heuristic"),
                           (dict(heuristic=4), UNSUPPORTED, "Heuristic"), (dict(heuristic=5), UNSUPPORTED, "Heuristic"), (dict(heuristic=6), UNSUPPORTED, "Heuristic"),
                           (dict(memory_mode=3, span=1), UNSUPPORTED, "BiWFA"), (dict(gap_ext1=0), INVALID, "positive"),
                           (dict(mismatch=-2), INVALID, "positive")):
        rc, status, _ = _wfa_call(_lib, ctx, params(**kw), pats, txts)
        assert rc == want, (kw, rc)
        assert word in _msg(_lib, ctx), (kw, _msg(_lib, ctx))
        assert (status == 99).all()      # nothing was written
    rc, _, _ = _wfa_call(_lib, ctx, params(), pats, txts, seqs_null=True)
    assert rc == INVALID and "null" in _msg(_lib, ctx)
    # the context is still good: the same call with valid parameters
    rc, status, score = _wfa_call(_lib, ctx, params(), pats, txts)
    assert rc == 0 and status.tolist() == [0, 0] and score.tolist() == [-2, -6]
    # an empty batch is fine
    rc, _, _ = _wfa_call(_lib, ctx, params(), [], [])
    assert rc == 0


def test_wfa_workspace_limit_is_an_error_not_a_crash(env):
    _lib, hmm, locus, synth, W, ctx = env
    rng = np.random.default_rng(1)
    a = bytes(rng.choice(list(b"ACGT"), 3000).tolist())
    b = bytes(rng.choice(list(b"ACGT"), 3000).tolist())   # unrelated 3-kb sequences, MemoryHigh: a large wavefront history
    p = _lib.WfaParams()
    _lib.lib().trgt_wfa_default_params(C.byref(p))
    p.metric, p.mismatch, p.gap_open1, p.gap_ext1, p.heuristic = 3, 2, 5, 1, 0
    assert _lib.lib().trgt_hip_set_workspace_limit(ctx.handle, C.c_uint64(1 << 16)) == 0
    rc, _, _ = _wfa_call(_lib, ctx, p, [a], [b])
    assert rc == NOMEM and "workspace" in _msg(_lib, ctx)
    assert _lib.lib().trgt_hip_set_workspace_limit(ctx.handle, C.c_uint64(8 << 30)) == 0
    rc, status, score = _wfa_call(_lib, ctx, p, [a], [b])
    assert rc == 0 and status[0] == 0 and score[0] < -1000


def test_hmm_batch_rejects_bad_jobs_and_motifs(env):
    _lib, H, locus, synth, W, ctx = env
    good = H.pack_hmm_batch([[b"CAG"]], [(0, b"CAGCAGCAG")])
    ok = H.hmm_batch(good, ctx)
    assert ok["counts"].tolist() == [3]
    bad = dict(good)
    bad["job_set"] = np.array([5], np.uint32)
    with pytest.raises(_lib.TrgtHipError, match="bad set"):
        H.hmm_batch(bad, ctx)
    empty = dict(good)
    empty["motif_off"] = np.array([0, 0], np.uint32)   # a motif of length 0
    with pytest.raises(_lib.TrgtHipError):
        H.hmm_batch(empty, ctx)
    again = H.hmm_batch(good, ctx)                      # context still usable
    assert again["counts"].tolist() == [3] and again["purity"][0] == 1.0


def test_locus_batch_rejects_short_flanks_and_small_allele_buffers(env):
    _lib, H, locus, synth, W, ctx = env
    b = synth.generate(8, first_locus=0)
    out = locus.run_batch(b, locus.Params(), ctx)
    assert int(out.n_alleles.sum()) == 16
    p = locus.Params()
    p.search_flank_len = 400                            # the batch carries 250-bp flanks
    with pytest.raises(_lib.TrgtHipError, match="flank shorter"):
        locus.run_batch(b, p, ctx)
    p.search_flank_len = 0
    with pytest.raises(_lib.TrgtHipError, match="flank_len"):
        locus.run_batch(b, p, ctx)
    small = locus.BatchOutputs(b)
    small.allele_cap[:] = 4                             # alleles are 30+ bp
    with pytest.raises(_lib.TrgtHipError, match="allele_cap"):
        locus.run_batch(b, locus.Params(), ctx, outputs=small)
    again = locus.run_batch(b, locus.Params(), ctx)     # and the context is still usable
    assert np.array_equal(again.allele_len, out.allele_len)
