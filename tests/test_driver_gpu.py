"""Several contexts on ONE GPU fed from the chunk queue of trgt_amd.driver (what bench.py measures `value` with): byte-identical to one
context working through the chunks in order, whether their flank-location stages overlap (the default) or not (TRGT_STAGE_LOCK)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config,n,chunk,k,lock", [(2, 1200, 250, 2, 0), (5, 160, 40, 2, 1), (4, 600, 128, 4, 0), (2, 2000, 200, 4, 0)])
def test_contexts_on_one_gpu_equal_one_context(oracle, config, n, chunk, k, lock):
    import torch
    from trgt_amd import _lib, locus, shard, synth
    from trgt_amd.driver import ChunkDriver, split_batch
    from test_locus_gpu import _compare
    b = synth.generate(n, first_locus=31000, config=config)
    chunks = split_batch(b, chunk)
    rd, fd = torch.from_numpy(b["read_blob"]).cuda(), torch.from_numpy(b["flank_blob"]).cuda()
    one = _lib.Context(0)
    ref = [locus.run_batch(c, ctx=one, flank_dev=fd, reads_dev=rd) for c in chunks]
    one.close()
    drv = ChunkDriver(devices=(0,) * k, context_factory=(lambda d: _lib.context_with_env(device=d, TRGT_STAGE_LOCK=1)) if lock else None)
    try:
        for rep in range(2):
            got = drv.run(chunks, per_chunk_kwargs=[dict(flank_dev=fd, reads_dev=rd) for _ in chunks])
            for c, g, r in zip(chunks, got, ref):
                assert shard.result_digest(g, c["n_loci"]) == shard.result_digest(r, c["n_loci"])
                for f in ("span_start", "span_end", "classification", "read_rank", "ci", "num_spanning"):
                    assert np.array_equal(getattr(g, f), getattr(r, f)), f
        assert sum(1 for v in drv.chunks_by_context if v > 0) >= 2  # more than one context took part
    finally:
        drv.close()
    _compare(oracle, locus, chunks[1], got[1], locus.Params(), range(0, chunks[1]["n_loci"], 7))
    whole = locus.run_batch(b, flank_dev=fd, reads_dev=rd)
    off = 0
    for c, g in zip(chunks, got):  # and the chunks concatenate to the unsplit batch
        nl = c["n_loci"]
        assert np.array_equal(g.n_alleles, whole.n_alleles[off:off + nl]) and np.array_equal(g.ci, whole.ci[4 * off:4 * (off + nl)])
        assert np.array_equal(g.purity.view(np.uint64), whole.purity[2 * off:2 * (off + nl)].view(np.uint64))
        off += nl


def test_native_pool_equals_one_context(oracle):
    # trgt_hip_pool / trgt_locus_batch_many: the same queue of batches inside the library (worker threads in C++), per-batch outputs
    import torch
    from trgt_amd import _lib, locus, shard, synth
    from trgt_amd.driver import split_batch
    b = synth.generate(1500, first_locus=47000, config=2)
    chunks = split_batch(b, 200)
    rd, fd = torch.from_numpy(b["read_blob"]).cuda(), torch.from_numpy(b["flank_blob"]).cuda()
    one = _lib.Context(0)
    ref = [locus.run_batch(c, ctx=one, flank_dev=fd, reads_dev=rd) for c in chunks]
    one.close()
    pool = _lib.Pool([0, 0, 0])
    try:
        for rep in range(2):
            got, ran = locus.run_many(pool, chunks, flank_dev=fd, reads_dev=rd)
            assert len(set(ran)) >= 2 and all(0 <= w < 3 for w in ran)
            for c, g, r in zip(chunks, got, ref):
                assert shard.result_digest(g, c["n_loci"]) == shard.result_digest(r, c["n_loci"])
        # per-context outputs: the last batch every context ran is what its entry holds
        outs, ran = locus.run_many(pool, [chunks[0]] * 7, flank_dev=fd, reads_dev=rd, out_per_context=True)
        for w in set(ran):
            assert shard.result_digest(outs[w], chunks[0]["n_loci"]) == shard.result_digest(ref[0], chunks[0]["n_loci"])
        # an error in one batch is reported with its index
        bad = dict(chunks[1])
        bad["lf_len"] = bad["lf_len"].copy()
        bad["lf_len"][3] = 10  # shorter than flank_len
        with pytest.raises(_lib.TrgtHipError, match="batch"):
            locus.run_many(pool, [chunks[0], bad, chunks[2]], flank_dev=fd, reads_dev=rd)
    finally:
        pool.close()


def test_native_pool_of_eight_contexts(oracle):
    # trgt_hip_pool_create with EIGHT entries -- the in-process form of the 8-GPU launch (one context per device of a node),
    # here eight contexts on the one GPU of the box ([0] * 8) -- so that the eight-worker path (eight threads, eight contexts with their
    # streams and buffer pools, one queue of batches) has run on hardware before the first real 8-GPU node sees it.  Host reads: every
    # context uploads its own batches one ahead (submit / wait), which is what a node-level driver would do per device.
    import torch
    from trgt_amd import _lib, locus, shard, synth
    from trgt_amd.driver import split_batch
    b = synth.generate(2400, first_locus=88000, config=4)
    chunks = split_batch(b, 150)
    fd = torch.from_numpy(b["flank_blob"]).cuda()
    one = _lib.Context(0)
    ref = [locus.run_batch(c, ctx=one, flank_dev=fd) for c in chunks]
    one.close()
    pool = _lib.Pool([0] * 8)
    try:
        assert len(pool.contexts) == 8
        for c in pool.contexts:  # (eight workspaces on one GPU: keep each small)
            c.check(_lib.lib().trgt_hip_set_workspace_limit(c.handle, 4 << 30))
        got, ran = locus.run_many(pool, chunks, flank_dev=fd)
        assert len(set(ran)) >= 4 and all(0 <= w < 8 for w in ran)
        for c, g, r in zip(chunks, got, ref):
            assert shard.result_digest(g, c["n_loci"]) == shard.result_digest(r, c["n_loci"])
    finally:
        pool.close()
