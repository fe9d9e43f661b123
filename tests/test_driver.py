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


def test_split_batch_keeps_read_metadata_and_encoding():
    """chunks must carry hp_tag / start_offset / end_offset / mismatch offsets (rebased) and read_encoding."""
    from trgt_amd.driver import split_batch
    from trgt_amd import locus
    rng = np.random.default_rng(3)
    loci = []
    for l in range(7):
        n = 3 + l
        loci.append(dict(left_flank=b"A" * 250, right_flank=b"C" * 250, tr=b"CAGCAG", motifs=[b"CAG"], ploidy=2,
                         reads=[b"ACGT" * (5 + i) for i in range(n)], hp_tag=[None, 1, 2][l % 3:] + [1] * (n - 3 + l % 3),
                         start_offset=[-10 * i for i in range(n)], end_offset=[7 * i for i in range(n)],
                         mismatch_offsets=[sorted(int(v) for v in rng.integers(-50, 50, size=i % 4)) for i in range(n)]))
    b = locus.pack(loci)
    b["read_encoding"] = 1  # (only the propagation is checked here: no library call)
    chunks = split_batch(b, 3)
    assert [c["n_loci"] for c in chunks] == [3, 3, 1]
    r = 0
    for c in chunks:
        assert c["read_encoding"] == 1
        for i in range(c["n_reads"]):
            assert c["hp_tag"][i] == b["hp_tag"][r] and c["start_offset"][i] == b["start_offset"][r] and c["end_offset"][i] == b["end_offset"][r]
            got = c["mismatch_offsets"][int(c["mismatch_off"][i]):int(c["mismatch_off"][i + 1])]
            want = b["mismatch_offsets"][int(b["mismatch_off"][r]):int(b["mismatch_off"][r + 1])]
            assert list(got) == list(want)
            r += 1
        assert int(c["mismatch_off"][0]) == 0 and len(c["mismatch_offsets"]) == int(c["mismatch_off"][-1]) + 1
    assert r == b["n_reads"]
