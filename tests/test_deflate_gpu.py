"""The device-side BGZF deflate (trgt_amd/csrc/deflate_dev.hip, trgt_deflate_blocks): every stream it produces must inflate -- with zlib,
the reference decoder -- to exactly the bytes it was given; data that does not fit the room is declined, not truncated.  And the writer
with trgt_writer_params.deflate_device: the same records in the spanning BAM as with zlib on host threads."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blocks():
    rng = np.random.default_rng(77)
    dna4 = lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes()   # packed bases of random sequence: hardly compressible
    quals = lambda n: np.repeat(rng.integers(20, 94, n // 7 + 1, dtype=np.uint8), rng.integers(1, 14, n // 7 + 1))[:n].astype(np.uint8).tobytes()
    repeat = lambda unit, n: (unit * (n // len(unit) + 1))[:n]
    out = [b"", b"A", b"ABC", b"ABCD", b"A" * 5, b"\x00" * 63, b"\xff" * 64, b"ab" * 33, bytes(range(256)) * 4, b"\x00" * 65536, b"\x90" * 65280,
           repeat(b"CAGCAGCAGCCG", 65280), repeat(b"\x12\x42\x81", 40000), quals(65280), quals(1000) + repeat(b"\x11\x22\x44\x88", 3000) + quals(333),
           (b"read_name/123/ccs\x00" + quals(900) + repeat(b"\x14\x28\x42", 600)) * 40, dna4(65280), dna4(100), dna4(4097), bytes(1) + dna4(2) + bytes(258) + dna4(2) + bytes(600)]
    for n in (255, 256, 257, 258, 259, 1019, 1020, 1021, 1023, 1024, 1025, 4095, 4096, 65279, 65535):
        out.append(repeat(b"ACGTTGCA" + bytes([n & 255]), n))
    return [o[:65536] for o in out]


def test_streams_inflate_to_their_input():
    from trgt_amd import _lib, writers
    ctx = _lib.Context(0)
    datas = _blocks() * 3   # (more blocks than one wave claims)
    got = writers.deflate_blocks(ctx, datas, cap=0xFF00)
    declined = 0
    for d, s in zip(datas, got):
        if s is None:
            declined += 1
            assert len(zlib.compress(d, 1)) > 0.8 * len(d) or len(d) * 9 // 8 + 2 > 0xFF00, len(d)   # only what does not compress is declined
            continue
        assert len(s) <= 0xFF00
        dec = zlib.decompressobj(-15)
        assert dec.decompress(s) + dec.flush() == d and dec.eof and not dec.unused_data, len(d)
    assert 0 < declined < len(datas) // 3
    # compressible data must come out smaller, about like zlib's fastest level
    rep = [d for d in datas if len(d) >= 40000 and len(zlib.compress(d, 1)) < len(d) // 3]
    assert rep
    for d in rep[:4]:
        s = writers.deflate_blocks(ctx, [d])[0]
        assert s is not None and len(s) < 3 * len(zlib.compress(d, 1)) + 2048, (len(s), len(zlib.compress(d, 1)))
    # deterministic
    again = writers.deflate_blocks(ctx, datas, cap=0xFF00)
    assert again == got
    # a tight room: declined, nothing written beyond it
    small = writers.deflate_blocks(ctx, [datas[11]] * 2, cap=64)
    assert small == [None, None] or all(s is None or len(s) <= 64 for s in small)
    ctx.close()


def test_writer_with_device_deflate_writes_the_same_records(tmp_path):
    from bamtools import read_bam_records
    from trgt_amd import ingest, locus, synth_bam, writers
    ds = synth_bam.write_dataset(str(tmp_path), n_loci=400, read_len=3000)
    rd = ingest.Reader(ds["bam"], ds["fasta"])
    b = rd.batch(ds["bed"], first_locus=0, max_loci=400, keep_native=True, copy=False, read_names=True, threads=4, keep_bam4=1)
    out = locus.run_batch(ingest.bam4_view(b))
    paths = {}
    for tag, dev in (("host", -1), ("dev", 0)):
        w = writers.Writer(rd, tmp_path / (tag + ".vcf"), tmp_path / (tag + ".bam"), deflate_device=dev)
        w.write(b, out)
        st = w.device_stats()   # (a caller who named a GPU can tell that it did the work)
        assert (st["device"] > 50 and st["declined"] == 0 and st["host"] == 0) if dev >= 0 else (st["device"] == 0 and st["host"] > 50), st
        w.close()
        paths[tag] = str(tmp_path / (tag + ".bam"))
    assert open(tmp_path / "host.vcf").read() == open(tmp_path / "dev.vcf").read()
    (ht, hr, hrec), (dt, dr, drec) = read_bam_records(paths["host"]), read_bam_records(paths["dev"])
    assert ht == dt and hr == dr and len(hrec) == len(drec) > 400 * 20
    assert hrec == drec
    assert os.path.getsize(paths["dev"]) < 2.5 * os.path.getsize(paths["host"])   # a fast level's ratio (against zlib level 6), not a stored file
    assert open(paths["dev"], "rb").read() != open(paths["host"], "rb").read()    # ... and not zlib's file: the device's blocks are in it
