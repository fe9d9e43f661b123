"""Two contexts on ONE GPU fed from the chunk queue of trgt_amd.driver: byte-identical to one context working through the chunks
in order (so that the first multi-GPU run is not also the first time two contexts coexist in a process)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config,n,chunk", [(2, 1200, 250), (5, 160, 40), (4, 600, 128)])
def test_two_contexts_one_gpu_equal_one_context(oracle, config, n, chunk):
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
    drv = ChunkDriver(devices=(0, 0))
    try:
        for rep in range(2):
            got = drv.run(chunks, per_chunk_kwargs=[dict(flank_dev=fd, reads_dev=rd) for _ in chunks])
            for c, g, r in zip(chunks, got, ref):
                assert shard.result_digest(g, c["n_loci"]) == shard.result_digest(r, c["n_loci"])
                for f in ("span_start", "span_end", "classification", "read_rank", "ci", "num_spanning"):
                    assert np.array_equal(getattr(g, f), getattr(r, f)), f
        assert min(drv.chunks_by_context) > 0  # both contexts took part
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
