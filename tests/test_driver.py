"""trgt_amd.driver: the chunk queue and the batch splitter (CPU: a stand-in run function; the GPU test runs real contexts)."""
import threading
import time

import numpy as np


def test_chunk_queue_is_dynamic_and_ordered():
    from trgt_amd.driver import ChunkDriver
    seen = []
    lock = threading.Lock()

    def run(ctx, chunk, params, kw):
        time.sleep(0.05 if chunk == 0 else 0.001)  # one slow chunk must not hold the others up
        with lock:
            seen.append((ctx, chunk))
        return chunk * 10

    d = ChunkDriver(devices=(0, 0, 1), params=object(), context_factory=lambda dev: "ctx%d_%d" % (dev, len(seen)), run_fn=run)
    d.contexts = ["a", "b", "c"]
    res = d.run(list(range(12)))
    assert res == [10 * i for i in range(12)]
    assert sum(d.chunks_by_context) == 12 and sorted(c for _, c in seen) == list(range(12))
    slow_ctx = [c for c, ch in seen if ch == 0][0]
    assert sum(1 for c, _ in seen if c == slow_ctx) < 6  # the worker stuck on the slow chunk took fewer of them


def test_worker_errors_surface():
    import pytest
    from trgt_amd.driver import ChunkDriver

    def run(ctx, chunk, params, kw):
        if chunk == 3:
            raise ValueError("boom")
        return chunk

    d = ChunkDriver(devices=(0, 0), params=object(), context_factory=lambda dev: dev, run_fn=run)
    with pytest.raises(ValueError, match="boom"):
        d.run(list(range(8)))


def test_split_batch_chunks_cover_the_batch(oracle):
    from trgt_amd import synth
    from trgt_amd.driver import split_batch
    b = synth.generate(37, first_locus=11, config=4)
    chunks = split_batch(b, 10)
    assert [c["n_loci"] for c in chunks] == [10, 10, 10, 7] and sum(c["n_reads"] for c in chunks) == b["n_reads"]
    # every locus of every chunk analyses to what it does in the whole batch (the oracle reads the chunk's own tables)
    whole = oracle.locus_records(b, 0, 37, 2)
    parts = []
    for c in chunks:
        parts += oracle.locus_records(c, 0, c["n_loci"], 2)
    assert parts == whole
