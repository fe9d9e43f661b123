"""Device-side read ingestion (trgt_ingest_params.ingest_device, trgt_amd/csrc/ingest_dev.hip: CRC-32 of the inflated BGZF blocks, the record
walk of extract_reads, HiFiRead::from_hts_rec, extract_snps_offset, get_meth and clip_to_region as kernels behind the device inflate)
against the host path of the same library, array by array, on: the reference's example data set, the hand-made records of
tests/test_ingest.py (soft clips, = / X / I / D runs, HP, MM / ML on both strands, secondary / supplementary / low-rq records, a second
contig), synthetic data sets of full-length reads, and the reference's own unit-test vectors for clip_to_region / extract_snps_offset /
get_meth written as BAM records (tests/golden/read_kats.json).  The host path itself is pinned by tests/test_ingest.py and
tests/test_read_helper_kats.py (CPU suite)."""
import json
import os
import re
import threading

import numpy as np
import pytest

from bamtools import write_bam, write_fasta

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
EX = os.path.join(GOLD, "example")
KATS = json.load(open(os.path.join(GOLD, "read_kats.json")))


def _same_batches(x, y):
    assert x["n_loci"] == y["n_loci"] and x["n_reads"] == y["n_reads"]
    for k in x:
        if k in ("read_blob_dev", "read_blob_device", "_native"):
            continue
        if isinstance(x[k], np.ndarray):
            assert x[k].dtype == y[k].dtype and x[k].shape == y[k].shape, k
            assert np.array_equal(x[k], y[k], equal_nan=True) if x[k].dtype.kind == "f" else np.array_equal(x[k], y[k]), k
        elif isinstance(x[k], list):
            assert x[k] == y[k], k


def _both(rd, bed, **kw):
    host = rd.batch(bed, **kw)
    before = rd.device_stats()
    dev = rd.batch(bed, ingest_device=0, **kw)
    after = rd.device_stats()
    assert after["calls"] == before["calls"] + 1
    return host, dev, after["fallbacks"] - before["fallbacks"], after


def test_example_data_set():
    from trgt_amd import ingest
    rd = ingest.Reader(os.path.join(EX, "sample.bam"), os.path.join(EX, "reference.fasta"))
    host, dev, fell_back, st = _both(rd, os.path.join(EX, "repeat.bed"), keep_bam4=1)
    assert not fell_back and st["blocks"] > 0 and host["n_reads"] == 33
    _same_batches(host, dev)
    assert dev["read_blob_device"] == 0 and dev["read_blob_dev"]


def test_hand_made_records_with_every_kind_of_field(tmp_path):
    from trgt_amd import ingest
    from test_ingest import _synthetic
    bam, fa, bed, recs, genome = _synthetic(tmp_path)
    rd = ingest.Reader(bam, fa)
    host, dev, fell_back, _ = _both(rd, bed, threads=2)
    assert not fell_back and host["n_reads"] >= 5 and host["has_meth"].any() and len(host["mismatch_offsets"]) > 0
    _same_batches(host, dev)
    host, dev, fell_back, _ = _both(rd, bed, min_read_qual=0.1, flank_len=100, keep_bam4=1)
    assert not fell_back
    _same_batches(host, dev)


def test_synthetic_data_set_and_the_reads_left_in_hbm(tmp_path):
    from trgt_amd import _lib, ingest, synth_bam
    ds = synth_bam.write_dataset(str(tmp_path / "ds"), n_loci=60, read_len=3000)
    rd = ingest.Reader(ds["bam"], ds["fasta"])
    host, dev, fell_back, st = _both(rd, ds["bed"], keep_bam4=1, keep_native=True)
    assert not fell_back and st["blocks_by_zlib"] == 0 and host["n_reads"] > 1000
    _same_batches(host, dev)
    # the ASCII reads the batch left in HBM, handed to trgt_locus_batch in place of an upload: same results as from the host blob
    from trgt_amd import locus, shard
    ctx = _lib.Context(0)
    want = locus.run_batch(host, locus.Params(), ctx)
    got = locus.run_batch(dev, locus.Params(), ctx, reads_dev=ingest.device_reads(dev))
    assert shard.result_digest(want, 60) == shard.result_digest(got, 60) and int((want.n_alleles == 2).sum()) > 50
    # a second, partial call on the same reader: slots and slabs are reused
    a, b, fell_back, _ = _both(rd, ds["bed"], first_locus=10, max_loci=25)
    assert not fell_back
    _same_batches(a, b)


def test_calls_from_several_threads_overlap_and_agree(tmp_path):
    from trgt_amd import ingest, synth_bam
    ds = synth_bam.write_dataset(str(tmp_path / "ds"), n_loci=80, read_len=2500)
    rd = ingest.Reader(ds["bam"], ds["fasta"])
    want = [rd.batch(ds["bed"], first_locus=a, max_loci=20) for a in range(0, 80, 20)]
    got, errs = [None] * 8, []

    def run(i):
        try:
            got[i] = rd.batch(ds["bed"], first_locus=20 * (i % 4), max_loci=20, ingest_device=0)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(i,)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(8):
        _same_batches(want[i % 4], got[i])
    assert rd.device_stats()["fallbacks"] == 0


def test_a_locus_deeper_than_the_reservoir_is_down_sampled_like_the_host_path(tmp_path):
    # more reads than 3 * max_depth: every further read replaces a random slot, by the same random stream as the host's StdRng(42)
    from trgt_amd import ingest
    from test_ingest import _synthetic
    bam, fa, bed, recs, genome = _synthetic(tmp_path, deep=200)
    rd = ingest.Reader(bam, fa)
    for depth in (10, 20, 50, 250):   # reservoirs of 30, 60, 150 (all smaller than the 201 reads over the second locus) and 750
        host, dev, fell_back, st = _both(rd, bed, max_depth=depth)
        assert not fell_back, st
        _same_batches(host, dev)
        assert int(host["n_reads_seen"][1]) == 201 and int(host["locus_read_begin"][2] - host["locus_read_begin"][1]) <= 3 * depth


def test_a_block_with_a_wrong_crc_is_refused(tmp_path):
    from trgt_amd import _lib, ingest, synth_bam
    ds = synth_bam.write_dataset(str(tmp_path / "ds"), n_loci=12, read_len=2000)
    raw = bytearray(open(ds["bam"], "rb").read())
    # the third block's footer: flip one bit of its CRC-32 (the inflated length still fits)
    off = 0
    for _ in range(2):
        off += (raw[off + 16] | (raw[off + 17] << 8)) + 1
    end = off + (raw[off + 16] | (raw[off + 17] << 8)) + 1
    raw[end - 8] ^= 1
    open(ds["bam"], "wb").write(bytes(raw))
    rd = ingest.Reader(ds["bam"], ds["fasta"])
    with pytest.raises(_lib.TrgtHipError, match="CRC-32"):
        rd.batch(ds["bed"], ingest_device=0)
    st = rd.device_stats()
    assert st["fallbacks"] == 1 and st["last_reason"] == 1
    with pytest.raises(_lib.TrgtHipError, match="CRC-32"):  # the host path refuses it as well (htslib does)
        rd.batch(ds["bed"])


def test_an_unusable_ingest_device_fails_the_call(tmp_path):
    from trgt_amd import _lib, ingest, synth_bam
    ds = synth_bam.write_dataset(str(tmp_path / "ds"), n_loci=12, read_len=2000)
    rd = ingest.Reader(ds["bam"], ds["fasta"])
    with pytest.raises(_lib.TrgtHipError, match="ingest_device 99"):
        rd.batch(ds["bed"], ingest_device=99)
    a, b = rd.batch(ds["bed"]), rd.batch(ds["bed"], ingest_device=0)
    _same_batches(a, b)


# ---- the reference's unit-test vectors as BAM records -----------------------------------------------------------------------------------
def _ops(cigar):
    return [(c, int(n)) for n, c in re.findall(r"(\d+)([MIDNSHP=X])", cigar)]


def _mm_for(bases, meth):
    """MM / ML tags (forward strand) that put meth[i] on the i-th CpG of `bases`"""
    cs = [i for i, ch in enumerate(bases) if ch == "C"]
    cpg = [k for k, i in enumerate(cs) if bases[i:i + 2] == "CG"]
    assert len(cpg) == len(meth)
    deltas, last = [], -1
    for k in cpg:
        deltas.append(k - last - 1)
        last = k
    return "C+m?," + ",".join(map(str, deltas)) + ";", list(meth)


def _one_read_data_set(tmp_path, tag, rec, region, flank_len, genome_len=12000):
    rng = np.random.default_rng(11)
    genome = "".join(rng.choice(list("ACGT"), genome_len))
    fa = str(tmp_path / (tag + ".fa"))
    write_fasta(fa, [("chr1", genome)])
    bed = str(tmp_path / (tag + ".bed"))
    open(bed, "w").write("chr1\t%d\t%d\tID=K;MOTIFS=CAG;STRUC=(CAG)n\n" % region)
    bam = str(tmp_path / (tag + ".bam"))
    write_bam(bam, [("chr1", genome_len)], [rec])
    return bam, fa, bed


@pytest.mark.parametrize("case", [c for c in KATS["clip_to_region"] if c["region_end"] - c["region_start"] >= 5],
                         ids=lambda c: "%s-%d-%d" % (c["test"], c["region_start"], c["region_end"]))
def test_clip_to_region_vectors_through_the_device(tmp_path, case):
    # clip_reads cuts at region -+ 2 * flank_len: flank_len 1 and the locus [start + 2, end - 2) put the cut where the vector has it
    from trgt_amd import ingest
    r = case["read"]
    mm, ml = _mm_for(r["bases"], r["meth"])
    rec = dict(name="kat", tid=0, pos=r["ref_pos"], cigar=_ops(r["cigar"]), seq=r["bases"], qual=[40] * len(r["bases"]),
               tags={"rq": ("f", 0.999), "MM": ("Z", mm), "ML": ("BC", ml)})
    bam, fa, bed = _one_read_data_set(tmp_path, "c", rec, (case["region_start"] + 2, case["region_end"] - 2), 1)
    rd = ingest.Reader(bam, fa)
    host, dev, fell_back, _ = _both(rd, bed, flank_len=1)
    assert not fell_back
    _same_batches(host, dev)
    exp = case["expected"]
    if exp is None:
        assert dev["n_reads"] == 0
        return
    assert dev["n_reads"] == 1
    assert bytes(dev["read_blob"][:int(dev["read_len"][0])]).decode() == exp["bases"]
    assert list(dev["meth"]) == exp["meth"] and int(dev["has_meth"][0]) == 1
    assert [(("MIDNSHP=X"[int(w) & 15]), int(w) >> 4) for w in dev["cigar"]] == _ops(exp["cigar"]) and int(dev["cigar_ref_pos"][0]) == exp["ref_pos"]
    assert bytes(dev["qual_blob"]) == bytes([40]) * len(exp["bases"])


def test_extract_snps_offset_vector_through_the_device(tmp_path):
    from trgt_amd import ingest
    c = KATS["extract_snps_offset"][0]
    ops = _ops(c["cigar"])
    qlen = sum(n for o, n in ops if o in "MIS=X")
    rng = np.random.default_rng(3)
    rec = dict(name="kat", tid=0, pos=c["ref_pos"], cigar=ops, seq="".join(rng.choice(list("ACGT"), qlen)), tags={"rq": ("f", 0.999)})
    ref_end = c["ref_pos"] + sum(n for o, n in ops if o in "MDN=X")
    bam, fa, bed = _one_read_data_set(tmp_path, "s", rec, (c["region"][0], c["region"][1]), 250, genome_len=ref_end + 2000)
    rd = ingest.Reader(bam, fa)
    host, dev, fell_back, _ = _both(rd, bed)
    assert not fell_back and dev["n_reads"] == 1
    _same_batches(host, dev)
    assert list(dev["mismatch_offsets"]) == c["expected"]


@pytest.mark.parametrize("case", KATS["get_meth"], ids=lambda c: c["test"])
def test_get_meth_vectors_through_the_device(tmp_path, case):
    from trgt_amd import ingest
    n = len(case["bases"])
    rec = dict(name="kat", tid=0, pos=1000, cigar=[("=", n)], seq=case["bases"], flag=16 if case["reverse"] else 0,
               tags={"rq": ("f", 0.999), "MM": ("Z", case["mm"]), "ML": ("BC", case["ml"])})
    bam, fa, bed = _one_read_data_set(tmp_path, "m", rec, (1000 + n // 2, 1000 + n // 2 + 1), 250)
    rd = ingest.Reader(bam, fa)
    host, dev, fell_back, _ = _both(rd, bed)
    assert not fell_back and dev["n_reads"] == 1
    _same_batches(host, dev)
    if case["expected"] is None:
        assert int(dev["has_meth"][0]) == 0 and len(dev["meth"]) == 0
    else:
        assert int(dev["has_meth"][0]) == 1 and list(dev["meth"]) == case["expected"]


def test_modification_strings_of_every_shape(tmp_path):
    """MM entries the parser must walk exactly like the host code: several entries, a numeric ChEBI code, two codes per entry (interleaved
    ML values), entries of other bases, '.' / '?' markers, a delta that runs off the read, too few ML values, both strands."""
    from trgt_amd import ingest
    rng = np.random.default_rng(17)
    genome = "".join(rng.choice(list("ACGT"), 9000))
    fa = str(tmp_path / "g.fa")
    write_fasta(fa, [("chr1", genome)])
    bed = str(tmp_path / "c.bed")
    open(bed, "w").write("chr1\t3000\t3060\tID=A;MOTIFS=CAG;STRUC=(CAG)n\n")
    recs = []
    shapes = ["C+m,0,1,0;", "C+76792,0;C+m,0,1;", "C+hm?,0,1,2;", "A+a,0,3;C+m.,1,1;", "C+m,2;C-m,0;G-m,1;C+m,1;", "C+m,0,100000;", "C+m,0,0,0,0,0,0,0,0;", "C+m;", "C+m,5", "N+n,1;C+m?,0;",
              "C+m," + ",".join(["0"] * 60) + ";"]
    for i, mm in enumerate(shapes):
        for flag in (0, 16):
            seq = "".join(rng.choice(list("ACGT"), 900))
            seq = seq[:300] + "CGCGACGTTCGA" * 20 + seq[540:]
            n_ml = mm.count(",") * (2 if "hm" in mm else 1)
            ml = [int(x) for x in rng.integers(1, 255, size=max(0, n_ml - (3 if i == 6 else 0)))]
            recs.append(dict(name="r%d_%d" % (i, flag), tid=0, pos=2600 + i, cigar=[("S", 3), ("=", 890), ("S", 7)], seq=seq, flag=flag,
                             tags={"rq": ("f", 0.999), "MM": ("Z", mm), "ML": ("BC", ml)}))
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "m.bam")
    write_bam(bam, [("chr1", 9000)], recs)
    rd = ingest.Reader(bam, fa)
    host, dev, fell_back, _ = _both(rd, bed)
    assert not fell_back and host["n_reads"] == len(recs) and host["has_meth"].sum() >= 8
    _same_batches(host, dev)
    host, dev, fell_back, _ = _both(rd, bed, flank_len=100)  # a narrower cut: the CpGs inside the clipped part only
    assert not fell_back
    _same_batches(host, dev)
