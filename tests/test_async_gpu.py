"""trgt_locus_batch_submit / _wait: the pipelined form of trgt_locus_batch (upload of batch k + 1 next to the compute of batch k)
must return exactly what the blocking call returns, batch after batch, and refuse tickets used out of order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    for f in ("span_start", "span_end", "n_alleles", "allele_len", "ci", "num_spanning", "classification", "read_rank", "n_spans"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert np.array_equal(a.purity.view(np.uint64), b.purity.view(np.uint64))
    from trgt_amd import shard
    assert shard.result_digest(a, len(a.n_alleles)) == shard.result_digest(b, len(b.n_alleles))


@pytest.mark.parametrize("config,n", [(2, 300), (5, 60)])
def test_pipelined_batches_equal_blocking_calls(oracle, config, n):
    import torch
    from trgt_amd import _lib, locus, synth
    from test_locus_gpu import _compare
    ctx = _lib.Context(0)
    batches = [synth.generate(n, first_locus=7000 + 1000 * i, config=config) for i in range(4)]
    pins = [torch.from_numpy(b["read_blob"]).pin_memory() for b in batches]
    fpins = [torch.from_numpy(b["flank_blob"]).pin_memory() for b in batches]
    ref = [locus.run_batch(b, ctx=ctx) for b in batches]
    tickets = [locus.submit_batch(batches[0], ctx=ctx, reads=pins[0], flank=fpins[0])]
    outs = []
    for k in range(4):
        if k + 1 < 4:
            tickets.append(locus.submit_batch(batches[k + 1], ctx=ctx, reads=pins[k + 1], flank=fpins[k + 1]))
        outs.append(tickets[k].wait())
    for k in range(4):
        _same(outs[k], ref[k])
    _compare(oracle, locus, batches[2], outs[2], locus.Params(), range(0, n, max(1, n // 25)))
    # pageable host memory and blobs already in HBM go through the same entry points
    t = locus.submit_batch(batches[1], ctx=ctx)
    _same(t.wait(), ref[1])
    t = locus.submit_batch(batches[3], ctx=ctx, reads=torch.from_numpy(batches[3]["read_blob"]).cuda(), flank=torch.from_numpy(batches[3]["flank_blob"]).cuda())
    _same(t.wait(), ref[3])
    ctx.close()


def test_ticket_rules():
    from trgt_amd import _lib, locus, synth
    ctx = _lib.Context(0)
    b = synth.generate(16, first_locus=5)
    t1 = locus.submit_batch(b, ctx=ctx)
    t2 = locus.submit_batch(b, ctx=ctx)
    with pytest.raises(_lib.TrgtHipError, match="outstanding"):
        locus.submit_batch(b, ctx=ctx)
    with pytest.raises(_lib.TrgtHipError, match="earlier"):
        t2.wait()
    o1 = t1.wait()
    o2 = t2.wait()
    _same(o1, o2)
    with pytest.raises(_lib.TrgtHipError, match="unknown ticket"):
        t2.wait()
    ctx.close()
