"""Native read ingestion (trgt_amd/csrc/ingest.hip: BGZF / BAI / FAI, extract_reads, HiFiRead::from_hts_rec, extract_snps_offset,
clip_to_region) against (a) the Python mirror tests/pyreads.py on the reference's example data set and (b) independent restatements
of the per-read rules on synthetic BAM files written by tests/bamtools.py.  No GPU involved."""
import os

import numpy as np
import pytest

from bamtools import write_bam, write_fasta

GOLD = os.path.join(os.path.dirname(__file__), "golden")
EX = os.path.join(GOLD, "example")


def _reads_of(b, l):
    a, e = int(b["locus_read_begin"][l]), int(b["locus_read_begin"][l + 1])
    return [bytes(b["read_blob"][int(b["read_off"][r]):int(b["read_off"][r]) + int(b["read_len"][r])]) for r in range(a, e)], range(a, e)


def test_example_data_set_matches_the_python_mirror():
    from trgt_amd import ingest
    import pyreads as reads
    b = ingest.Reader(os.path.join(EX, "sample.bam"), os.path.join(EX, "reference.fasta")).batch(os.path.join(EX, "repeat.bed"))
    genome = reads.read_fasta(os.path.join(EX, "reference.fasta"))
    loci = reads.read_catalog(os.path.join(EX, "repeat.bed"), genome)
    records = reads.read_bam(os.path.join(EX, "sample.bam"))
    assert b["n_loci"] == len(loci) == 1 and b["id"] == [loci[0].id] and b["contig"] == [loci[0].contig]
    L = reads.locus_inputs(loci[0], records)
    got, _ = _reads_of(b, 0)
    assert got == L["reads"] and len(got) == 33
    assert np.array_equal(b["read_qual"], np.array(L["read_qual"], np.float64))
    assert bytes(b["flank_blob"][int(b["lf_off"][0]):int(b["lf_off"][0]) + 250]) == L["left_flank"]
    assert bytes(b["flank_blob"][int(b["rf_off"][0]):int(b["rf_off"][0]) + 250]) == L["right_flank"]
    assert bytes(b["tr_blob"][:int(b["tr_len"][0])]) == L["tr"]
    assert bytes(b["motif_blob"]) == b"CAG" and b["struc"] == [loci[0].struc]
    assert int(b["n_reads_seen"][0]) == 33 and int(b["n_quality_filtered"][0]) == 0 and not b["has_meth"].any() and (b["hp_tag"] == -1).all()


def _comp(s):
    return s.translate(str.maketrans("ACGT", "TGCA"))[::-1]


def _synthetic(tmp_path, deep=0):
    rng = np.random.default_rng(5)
    genome = "".join(rng.choice(list("ACGT"), 6000))
    fa = str(tmp_path / "g.fa")
    write_fasta(fa, [("chr1", genome), ("chr2", genome[::-1])])
    bed = str(tmp_path / "cat.bed")
    open(bed, "w").write("chr1\t2000\t2060\tID=L1;MOTIFS=CAG,CCG;STRUC=(CAG)n(CCG)n\n\nchr1\t4000\t4030\tID=L2;MOTIFS=A;STRUC=(A)n\n")
    recs = []

    def add(name, pos, cigar, flag=0, rq=0.999, hp=None, mm=None, ml=None, tid=0):
        qlen = sum(n for c, n in cigar if c in "MIS=X")
        seq = "".join(rng.choice(list("ACGT"), qlen))
        if mm == "auto":  # a 5mC call on every second CpG (original strand), probabilities 10, 20, ...
            orig = _comp(seq) if flag & 16 else seq
            cs = [i for i, ch in enumerate(orig) if ch == "C"]
            cpg = [k for k, i in enumerate(cs) if orig[i:i + 2] == "CG"][::2]
            deltas, last = [], -1
            for k in cpg:
                deltas.append(k - last - 1)
                last = k
            mm, ml = "C+m?," + ",".join(map(str, deltas)) + ";" if deltas else "C+m?;", [(10 * (i + 1)) % 256 for i in range(len(deltas))]
        tags = {}
        if rq is not None:
            tags["rq"] = ("f", rq)
        if hp is not None:
            tags["HP"] = ("C", hp)
        if mm is not None:
            tags["MM"] = ("Z", mm)
            tags["ML"] = ("BC", ml)
        recs.append(dict(name=name, tid=tid, pos=pos, cigar=cigar, seq=seq, flag=flag, tags=tags, qual=[int(x) for x in rng.integers(2, 60, qlen)]))

    add("spans_all", 1400, [("S", 7), ("=", 500), ("X", 2), ("=", 98), ("I", 5), ("=", 60), ("D", 3), ("X", 1), ("=", 400), ("S", 4)], hp=1, mm="auto")
    add("rev_meth", 1450, [("=", 300), ("X", 1), ("=", 249), ("X", 3), ("=", 700)], flag=16, hp=2, mm="auto")
    add("starts_inside", 2030, [("M", 600)])
    add("ends_inside", 1300, [("M", 720)], rq=None)
    add("low_rq", 1500, [("M", 900)], rq=0.5)
    add("secondary", 1500, [("M", 900)], flag=256)
    add("supplementary", 1500, [("M", 900)], flag=2048)
    add("far_left", 100, [("M", 1000)])
    add("touches_region_end", 2309, [("M", 50)])
    add("just_outside", 2310, [("M", 50)])
    add("other_contig", 1500, [("M", 900)], tid=1)
    add("locus2", 3700, [("=", 290), ("X", 1), ("=", 9), ("D", 30), ("=", 400)], hp=1)
    for i in range(deep):
        add("deep%04d" % i, 3800 + (i % 50), [("M", 500)])
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    bam = str(tmp_path / "s.bam")
    write_bam(bam, [("chr1", 6000), ("chr2", 6000)], recs)
    return bam, fa, bed, recs, genome


def _expected_snps(rec, start, end):
    out, ref = [], rec["pos"]
    for c, n in rec["cigar"]:
        if c == "X" and not (start <= ref <= end):
            d = ref - start if ref < start else ref - end
            out += [d + i for i in range(n)]
            ref += n
        elif c in "MX=DN":
            ref += n
    return out


def _expected_meth(rec):
    """get_meth (read.rs:55-96) restated from the MM / ML definition: per CpG of the stored sequence"""
    tags = rec["tags"]
    if "MM" not in tags:
        return None
    seq, rev = rec["seq"], bool(rec["flag"] & 16)
    orig = _comp(seq) if rev else seq
    deltas = [int(x) for x in tags["MM"][1].split(";")[0].split(",")[1:]]
    cs = [i for i, ch in enumerate(orig) if ch == "C"]
    calls, k = {}, -1
    for d, q in zip(deltas, tags["ML"][1]):
        k += d + 1
        calls[cs[k]] = q
    cpgs_orig = [i for i in range(len(orig) - 1) if orig[i:i + 2] == "CG"]
    vals = [calls.get(i, 0) for i in cpgs_orig]
    if not any(i in calls for i in cpgs_orig):
        return None
    return vals  # get_meth reverses the reverse-strand list back into ORIGINAL order: vals is already in original order


def test_synthetic_bam_records(tmp_path):
    from trgt_amd import ingest
    import pyreads as reads
    bam, fa, bed, recs, genome = _synthetic(tmp_path)
    b = ingest.Reader(bam, fa).batch(bed, threads=2)
    assert b["n_loci"] == 2 and b["id"] == ["L1", "L2"] and list(b["set_motif_begin"]) == [0, 2, 3]
    assert bytes(b["tr_blob"][:60]).decode() == genome[2000:2060] and bytes(b["flank_blob"][:250]).decode() == genome[1750:2000]
    by_name = {r["name"]: r for r in recs}
    got, idx = _reads_of(b, 0)
    names = [b["read_name"][r] for r in idx]
    # region +- flank_len = [1750, 2310): file order, secondary / supplementary dropped, the low-quality one counted
    assert names == ["ends_inside", "spans_all", "rev_meth", "starts_inside", "touches_region_end"]
    assert int(b["n_quality_filtered"][0]) == 1 and int(b["n_reads_seen"][0]) == 5
    for r, name, bases in zip(idx, names, got):
        rec = by_name[name]
        mirror = reads.BamRecord(name, "chr1", rec["pos"], rec["flag"], [("MIDNSHP=X".index(c), n) for c, n in rec["cigar"]], rec["seq"], None)
        assert bases == reads.clip_to_region(mirror, (2000 - 500, 2060 + 500)), name
        q0 = rec["seq"].find(bases.decode()) if bases else 0
        assert bytes(b["qual_blob"][int(b["read_off"][r]):int(b["read_off"][r]) + len(bases)]) == bytes(rec["qual"][q0:q0 + len(bases)]), name
        ref_end = rec["pos"] + sum(n for c, n in rec["cigar"] if c in "MDN=X")
        assert int(b["start_offset"][r]) == rec["pos"] - 2000 and int(b["end_offset"][r]) == ref_end - 2060
        assert list(b["mismatch_offsets"][int(b["mismatch_off"][r]):int(b["mismatch_off"][r + 1])]) == _expected_snps(rec, 2000, 2060), name
        assert int(b["hp_tag"][r]) == (rec["tags"]["HP"][1] if "HP" in rec["tags"] else -1)
        assert bool(b["is_reverse"][r]) == bool(rec["flag"] & 16)
        rq = rec["tags"]["rq"][1] if "rq" in rec["tags"] else None
        assert (np.isnan(b["read_qual"][r]) if rq is None else b["read_qual"][r] == np.float32(rq))
        exp = _expected_meth(rec)
        if exp is None:
            assert not b["has_meth"][r]
        else:  # clip: the CpGs whose C lies inside the clipped bases (stored orientation)
            seq = rec["seq"]
            stored = exp[::-1] if rec["flag"] & 16 else exp   # get_meth's list is in original-strand order ...
            stored = stored[::-1] if rec["flag"] & 16 else stored  # ... which clip_to_region walks against the STORED bases as is
            cpg = [i for i in range(len(seq) - 1) if seq[i:i + 2] == "CG"]
            keep = [v for i, v in zip(cpg, stored) if q0 <= i < q0 + len(bases)]
            assert b["has_meth"][r] and list(b["meth"][int(b["meth_off"][r]):int(b["meth_off"][r + 1])]) == keep, name
    got2, idx2 = _reads_of(b, 1)
    assert [b["read_name"][r] for r in idx2] == ["locus2"] and int(b["hp_tag"][idx2[0]]) == 1
    r2 = idx2[0]
    assert list(b["mismatch_offsets"][int(b["mismatch_off"][r2]):int(b["mismatch_off"][r2 + 1])]) == _expected_snps(by_name["locus2"], 4000, 4030)


def test_reservoir_caps_deep_loci(tmp_path):
    from trgt_amd import ingest
    bam, fa, bed, recs, _ = _synthetic(tmp_path, deep=400)
    rd = ingest.Reader(bam, fa)
    b = rd.batch(bed, max_depth=40, threads=1)
    _, idx = _reads_of(b, 1)
    assert int(b["n_reads_seen"][1]) == 401 and len(idx) == 120  # 3 * max_depth kept of the reads that passed the filters
    names = [b["read_name"][r] for r in idx]
    assert len(set(names)) == 120 and any(n > "deep0119" for n in names if n.startswith("deep"))  # later reads replaced earlier ones
    again = rd.batch(bed, max_depth=40, threads=4)
    assert [again["read_name"][r] for r in _reads_of(again, 1)[1]] == names  # a fixed seed: the same sample every time


def test_reservoir_slots_are_rebuilt_from_scratch(tmp_path):
    # round 6 (found by tests/tools/ingest_fuzz.py against the device path): a read that replaces a reservoir slot and has no rq tag must
    # come out with read_qual None (NaN), not with the value of the read it replaced -- HiFiRead::from_hts_rec builds a new read every time
    from trgt_amd import ingest
    rng = np.random.default_rng(9)
    genome = "".join(rng.choice(list("ACGT"), 6000))
    fa = str(tmp_path / "g.fa")
    write_fasta(fa, [("chr1", genome)])
    bed = str(tmp_path / "c.bed")
    open(bed, "w").write("chr1\t3000\t3030\tID=L;MOTIFS=CAG;STRUC=(CAG)n\n")
    recs = [dict(name="r%03d" % i, tid=0, pos=2500 + (i % 40), cigar=[("M", 900)], seq="".join(rng.choice(list("ACGT"), 900)),
                 tags={"rq": ("f", 0.999)} if i % 3 else {}) for i in range(120)]
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "r.bam")
    write_bam(bam, [("chr1", 6000)], recs)
    b = ingest.Reader(bam, fa).batch(bed, max_depth=5, min_read_qual=0.5)
    assert int(b["n_reads_seen"][0]) == 120 and b["n_reads"] == 15
    tagged = {r["name"]: bool(r["tags"]) for r in recs}
    for name, rq in zip(b["read_name"], b["read_qual"]):
        assert np.isnan(rq) != tagged[name], (name, rq)
    assert np.isnan(b["read_qual"]).any() and (~np.isnan(b["read_qual"])).any()


def test_errors_are_reported(tmp_path):
    from trgt_amd import _lib, ingest
    with pytest.raises(_lib.TrgtHipError):
        ingest.Reader(str(tmp_path / "missing.bam"), os.path.join(EX, "reference.fasta"))
    rd = ingest.Reader(os.path.join(EX, "sample.bam"), os.path.join(EX, "reference.fasta"))
    bad = str(tmp_path / "bad.bed")
    # a catalog line that gives no locus is reported and skipped, as stream_loci_into_channel does (locus.rs:93-137): the good line survives
    open(bad, "w").write("chrA\t10001\t10061\tID=TR1;MOTIFS=CAG\nchrA\t100\t161\tID=TR1;MOTIFS=CAG;STRUC=(CAG)n\nchrA\t10001\t10061\tID=TR1;MOTIFS=CAG;STRUC=(CAG)n\n")
    b = rd.batch(bad)
    assert b["n_loci"] == 1 and b["id"] == ["TR1"] and len(b["skipped"]) == 2
    assert b["skipped"][0] == "Error at BED line 1: STRUC field missing" and b["skipped"][1].startswith("Error at BED line 2: Region start '100' with flank length '250' underflows")
    # Locus::new (locus.rs:50-60) checks the region bounds BEFORE it decodes the info field: a line with both problems reports the bounds
    open(bad, "w").write("chrA\t100\t161\tID=TR1;MOTIFS=CAG\nchrZ\t10001\t10061\tID=TR1;MOTIFS\n")
    b = rd.batch(bad)
    assert b["n_loci"] == 0 and len(b["skipped"]) == 2
    assert b["skipped"][0].startswith("Error at BED line 1: Region start '100' with flank length '250' underflows")
    assert b["skipped"][1] == "Error at BED line 2: FASTA reference does not contain chromosome 'chrZ' in BED file"
    with pytest.raises(_lib.TrgtHipError):
        rd.batch(str(tmp_path / "missing.bed"))


def test_truncated_and_corrupt_files_give_errors_not_crashes(tmp_path):
    import shutil
    from trgt_amd import _lib, ingest
    fa = os.path.join(EX, "reference.fasta")
    data = open(os.path.join(EX, "sample.bam"), "rb").read()
    outcomes = []
    for tag, blob in [("cut90", data[:int(len(data) * 0.9)]), ("cut50", data[:len(data) // 2]), ("cut1", data[:len(data) // 100]),
                      ("flip", data[:60000] + bytes(b ^ 0x5A for b in data[60000:60200]) + data[60200:])]:
        bam = str(tmp_path / (tag + ".bam"))
        open(bam, "wb").write(blob)
        shutil.copy(os.path.join(EX, "sample.bam.bai"), bam + ".bai")
        try:
            b = ingest.Reader(bam, fa).batch(os.path.join(EX, "repeat.bed"))
            outcomes.append((tag, int(b["n_reads"])))
        except _lib.TrgtHipError as e:
            outcomes.append((tag, str(e)))
    assert all(isinstance(v, str) or 0 <= v <= 33 for _, v in outcomes), outcomes
    assert any(isinstance(v, str) for _, v in outcomes), outcomes  # at least the badly damaged files are reported
    bam = str(tmp_path / "badidx.bam")
    shutil.copy(os.path.join(EX, "sample.bam"), bam)
    open(bam + ".bai", "wb").write(b"BAI\1" + b"\xff" * 40)
    with pytest.raises(_lib.TrgtHipError):
        ingest.Reader(bam, fa)


def test_keep_bam4_holds_the_same_bases_as_the_ascii_reads():
    # trgt_ingest_params.keep_bam4: the clipped reads once more as BAM 4-bit codes (TRGT_READS_BAM4), byte-aligned per read
    from trgt_amd import ingest
    rd = ingest.Reader(os.path.join(EX, "sample.bam"), os.path.join(EX, "reference.fasta"))
    b = rd.batch(os.path.join(EX, "repeat.bed"), keep_bam4=1)
    plain = rd.batch(os.path.join(EX, "repeat.bed"))
    assert "read_bam4" not in plain
    letters = "=ACMGRSVTWYHKDBN"
    nr = int(b["n_reads"])
    assert nr > 0 and np.array_equal(b["read_blob"], plain["read_blob"])
    o = 0
    for r in range(nr):
        n, off = int(b["read_len"][r]), int(b["read_bam4_off"][r])
        assert off == o
        packed = b["read_bam4"][off:off + (n + 1) // 2]
        dec = "".join(letters[x >> 4] + letters[x & 15] for x in packed)[:n]
        assert dec == bytes(b["read_blob"][int(b["read_off"][r]):int(b["read_off"][r]) + n]).decode()
        o += (n + 1) // 2
    v = ingest.bam4_view(b)
    assert v["read_encoding"] == 1 and v["read_blob"] is b["read_bam4"] and v["read_len"] is b["read_len"]
    rd.close()


def test_synthetic_dataset_ingests_the_same_whatever_the_threads(tmp_path):
    """trgt_amd/synth_bam.py (the end-to-end leg of bench.py): every locus gets its 30 reads, and the batch does not depend on the number
    of ingestion threads, on the block cache (runs of loci per worker) or on copy=False (views of the native batch)."""
    from trgt_amd import ingest, synth_bam
    ds = synth_bam.write_dataset(str(tmp_path / "ds"), n_loci=48, read_len=1500)
    rd = ingest.Reader(ds["bam"], ds["fasta"])
    a = rd.batch(ds["bed"], threads=1, keep_bam4=1)
    assert a["n_loci"] == 48 and a["n_reads"] == 48 * 30 and not a["skipped"] and (a["n_reads_seen"] == 30).all()
    b = rd.batch(ds["bed"], threads=5, keep_bam4=1, keep_native=True, copy=False, read_names=False)
    for k, v in a.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, b[k]), k
    assert b["read_name"] is None and a["id"] == b["id"]
    # the reads carry the alleles the data set was made from: the bases between the flanks of read 0 of locus 0
    lf = bytes(a["flank_blob"][int(a["lf_off"][0]):int(a["lf_off"][0]) + 250])
    r0 = bytes(a["read_blob"][int(a["read_off"][0]):int(a["read_off"][0]) + int(a["read_len"][0])])
    assert r0.count(lf[-40:]) == 1 or lf[-40:] not in r0  # (a substitution may sit in the flank)
    # the 4-bit copy is the ASCII one packed
    from trgt_amd import _lib
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    n0 = int(a["read_len"][0])
    packed = a["read_bam4"][int(a["read_bam4_off"][0]):int(a["read_bam4_off"][0]) + (n0 + 1) // 2]
    assert [int(x) >> 4 for x in packed[:8]] == [code[chr(c)] for c in r0[0:16:2]]
