"""Reads handed over as BAM 4-bit codes (TRGT_READS_BAM4, include/trgt_hip.h): the packing helper against a plain Python packer (CPU),
and on the GPU the same results, field by field, as the ASCII form of the batch -- blocking call, reads already in HBM, submit / wait
and the context pool."""
import ctypes as C

import numpy as np
import pytest

LETTERS = "=ACMGRSVTWYHKDBN"


def _pack_py(seq: bytes) -> bytes:
    code = {ord(ch): k for k, ch in enumerate(LETTERS)}
    code.update({ord(ch.lower()): k for k, ch in enumerate(LETTERS)})
    out = bytearray((len(seq) + 1) // 2)
    for i, b in enumerate(seq):
        k = code.get(b, 15)
        out[i // 2] |= (k << 4) if i % 2 == 0 else k
    return bytes(out)


def test_pack_helper_matches_python_packer():
    from trgt_amd import _lib
    rng = np.random.default_rng(11)
    reads = [b"", b"A", b"AC", b"ACG", b"NNNNN", b"acgtn", b"=ACMGRSVTWYHKDBN", b"AXZ?C"]  # unknown letters become N (htslib)
    reads += [bytes(rng.choice(list(b"ACGTN"), size=int(n))) for n in rng.integers(0, 400, 50)]
    # the reads sit in the ASCII blob with gaps and out of order
    order = rng.permutation(len(reads))
    blob, off = bytearray(), np.zeros(len(reads), np.uint64)
    for r in order:
        blob += b"#" * int(rng.integers(0, 5))
        off[r] = len(blob)
        blob += reads[r]
    blob += b"#"
    ascii_blob = np.frombuffer(bytes(blob), np.uint8)
    lens = np.array([len(r) for r in reads], np.uint32)
    total = int(((lens.astype(np.int64) + 1) // 2).sum())
    packed, poff = np.full(total + 3, 0xEE, np.uint8), np.zeros(len(reads), np.uint64)
    got = _lib.lib().trgt_reads_pack_bam4(_lib.ptr(ascii_blob), len(reads), _lib.ptr(off), _lib.ptr(lens), _lib.ptr(packed), _lib.ptr(poff))
    assert got == total
    assert bytes(packed[total:]) == b"\xEE" * 3  # nothing written past the end
    o = 0
    for r, seq in enumerate(reads):
        assert int(poff[r]) == o
        assert bytes(packed[o:o + (len(seq) + 1) // 2]) == _pack_py(seq), r
        o += (len(seq) + 1) // 2
    assert _lib.lib().trgt_reads_pack_bam4(None, 3, None, None, None, None) < 0


def _same(a, b):
    for f in ("span_start", "span_end", "n_alleles", "allele_len", "allele_blob", "ci", "num_spanning", "classification", "read_rank", "n_spans",
              "spans3", "motif_counts", "gt_size", "flipped"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert np.array_equal(a.purity.view(np.uint64), b.purity.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("config,n", [(2, 400), (4, 200), (5, 60), (3, 12)])
def test_packed_reads_give_the_results_of_ascii_reads(oracle, config, n):
    import torch
    from trgt_amd import _lib, locus, synth
    from test_locus_gpu import _compare
    ctx = _lib.Context(0)
    b = synth.generate_cfg3(n, first_locus=3) if config == 3 else synth.generate(n, first_locus=4100, config=config)
    ref = locus.run_batch(b, ctx=ctx)
    pk = locus.pack_bam4(b)
    assert pk["read_blob"].nbytes * 2 <= b["read_blob"].nbytes + 2 * len(b["read_len"]) + 2
    out = locus.run_batch(pk, ctx=ctx)                                   # pageable host memory
    _same(out, ref)
    _compare(oracle, locus, b, out, locus.Params(), range(0, n, max(1, n // 20)))
    dev = torch.from_numpy(pk["read_blob"]).cuda()
    _same(locus.run_batch(pk, ctx=ctx, reads_dev=dev), ref)              # packed blob already in HBM
    pinned = locus.pack_bam4(b, pinned=True)
    t1 = locus.submit_batch(pinned, ctx=ctx)                             # pipelined form, pinned host memory
    t2 = locus.submit_batch(pk, ctx=ctx)
    _same(t1.wait(), ref)
    _same(t2.wait(), ref)
    ctx.close()


@pytest.mark.gpu
def test_packed_reads_through_the_pool_and_next_to_ascii_batches():
    from trgt_amd import _lib, locus, synth
    batches = [synth.generate(150, first_locus=900 + 300 * i, config=2 if i % 2 == 0 else 5) for i in range(6)]
    ctx = _lib.Context(0)
    ref = [locus.run_batch(b, ctx=ctx) for b in batches]
    ctx.close()
    mixed = [locus.pack_bam4(b) if i % 3 != 2 else b for i, b in enumerate(batches)]
    pool = _lib.Pool([0, 0, 0])
    outs, ran = locus.run_many(pool, mixed)
    for o, r in zip(outs, ref):
        _same(o, r)
    pool.close()


@pytest.mark.gpu
def test_unknown_read_encoding_is_refused():
    from trgt_amd import _lib, locus, synth
    ctx = _lib.Context(0)
    b = dict(synth.generate(8, first_locus=1))
    b["read_encoding"] = 7
    with pytest.raises(_lib.TrgtHipError, match="read_encoding"):
        locus.run_batch(b, ctx=ctx)
    ctx.close()
