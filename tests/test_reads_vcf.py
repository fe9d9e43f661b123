"""Host-side steps either side of the GPU path: catalog + reference + BAM -> clipped reads
(tests/pyreads.py, mirror of tr.rs:186-196, 262-361 and clip_region.rs) and LocusResult -> VCF record (tests/pyvcf.py, mirror of
write_vcf.rs:95-397).  Inputs are the reference's own example data set (tests/golden/example/ = example/{reference.fasta,
repeat.bed, sample.bam}); the expected record is the one the reference documents for it (docs/tutorial.md:43-46)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
EX = os.path.join(GOLD, "example")
# docs/tutorial.md:45 (`bcftools view --no-header sample.vcf.gz | head -n 1`), tab-separated
TUTORIAL_RECORD = "\t".join([
    "chrA", "10001", ".", "C" + "CAG" * 20, "C" + "CAG" * 11, ".", ".", "TRID=TR1;END=10061;MOTIFS=CAG;STRUC=<TR>",
    "GT:AL:ALLR:SD:MC:MS:AP:AM", "1/1:33,33:30-39,33-33:15,14:11,11:0(0-33),0(0-33):1.000000,1.000000:.,."])


def _example():
    import pyreads as reads
    genome = reads.read_fasta(os.path.join(EX, "reference.fasta"))
    loci = reads.read_catalog(os.path.join(EX, "repeat.bed"), genome)
    records = reads.read_bam(os.path.join(EX, "sample.bam"))
    return reads, loci, records


def test_bam_ingestion_reproduces_the_committed_fixture():
    reads, loci, records = _example()
    assert len(loci) == 1 and loci[0].id == "TR1" and (loci[0].contig, loci[0].start, loci[0].end) == ("chrA", 10001, 10061)
    fx = json.load(open(os.path.join(GOLD, "example_e1_reads.json")))["loci"][0]
    L = reads.locus_inputs(loci[0], records)
    assert L["left_flank"].decode() == fx["left_flank"] and L["right_flank"].decode() == fx["right_flank"] and L["tr"].decode() == fx["tr"]
    assert [r.decode() for r in L["reads"]] == fx["reads"]
    assert all(q is not None and q >= 0.98 for q in L["read_qual"])


def test_clip_cigar_cases():
    from pyreads import clip_cigar
    M, I, D, S = 0, 1, 2, 4
    # read [100, 160) on the reference, 5 soft-clipped bases in front
    ops = [(S, 5), (M, 20), (I, 3), (M, 10), (D, 4), (M, 26)]
    assert clip_cigar(100, ops, (0, 1000)) == (100, 0, ops)                     # region covers everything: soft clip kept
    assert clip_cigar(100, ops, (110, 125)) == (110, 15, [(M, 10), (I, 3), (M, 5)])
    assert clip_cigar(100, ops, (131, 140)) == (131, 38, [(D, 3), (M, 6)])      # starts inside the deletion
    assert clip_cigar(100, ops, (160, 170)) is None and clip_cigar(100, ops, (0, 100)) is None


def test_set_gt_and_record_shapes():
    import pyreads as reads
    import pyvcf as vcf
    from trgt_amd.hmm import Annotation, Span
    from trgt_amd.locus import Allele, LocusResult
    loc = reads.Locus("X", "chr1", 101, 110, b"ACGTT", b"CAGCAGCAG", b"GGGGG", ["CAG"], "(CAG)n")
    ann = lambda n: Annotation([Span(0, 0, 3 * n)] if n else None, [n], 1.0 if n else float("nan"))
    al = lambda n: Allele(b"CAG" * n, ann(n), (3 * n, 3 * n), 5)
    rec = lambda g: vcf.vcf_record(loc, LocusResult(g, [], [], [])).split("\t")
    r = rec([al(3), al(3)])                       # homozygous reference
    assert r[3] == "TCAGCAGCAG" and r[4] == "." and r[9].startswith("0/0:9,9:")
    r = rec([al(3), al(5)])                       # reference first
    assert r[4] == "T" + "CAG" * 5 and r[9].startswith("0/1:9,15:9-9,15-15:5,5:3,5:0(0-9),0(0-15):1.000000,1.000000:.,.")
    r = rec([al(4), al(4)])                       # homozygous alternative: one ALT, 1/1
    assert r[4] == "T" + "CAG" * 4 and r[9].startswith("1/1:")
    r = rec([al(2), al(5)])                       # two alternatives
    assert r[4] == "T" + "CAG" * 2 + ",T" + "CAG" * 5 and r[9].startswith("1/2:")
    r = rec([al(0), al(3)])                       # empty allele: MS '.', AP '.'
    assert r[4] == "T" and r[9] == "1/0:0,9:0-0,9-9:5,5:0,3:.,0(0-9):.,1.000000:.,."
    r = rec([al(4)])                              # haploid
    assert r[9].startswith("1:12:12-12:5:4:")
    r = rec([])                                   # LocusResult::empty
    assert r[3] == "TCAGCAGCAG" and r[4] == "." and r[9] == ".:.:.:.:.:.:.:."
    assert r[:3] == ["chr1", "101", "."] and r[7] == "TRID=X;END=110;MOTIFS=CAG;STRUC=(CAG)n" and r[8] == "GT:AL:ALLR:SD:MC:MS:AP:AM"


def test_oracle_result_renders_as_the_tutorial_record(oracle):
    import pyvcf as vcf
    from trgt_amd.hmm import Annotation, Span
    from trgt_amd.locus import Allele, LocusResult
    reads, loci, records = _example()
    L = reads.locus_inputs(loci[0], records)
    r = oracle.locus_analyze(L["left_flank"], L["right_flank"], L["tr"], L["motifs"], L["reads"])
    geno = []
    for a in range(r["n_alleles"]):
        ms = r["MS"].split(",")[a]
        labels = None if ms == "." else [Span(int(x.split("(")[0]), int(x.split("(")[1].split("-")[0]), int(x.split("-")[1][:-1])) for x in ms.split("_")]
        counts = [int(v) for v in r["MC"].split(",")[a].split("_")]
        ap = r["AP"].split(",")[a]
        geno.append(Allele(r["alleles"][a].encode(), Annotation(labels, counts, float("nan") if ap == "." else float(ap)),
                           (int(r["gt_ci"][a][0]), int(r["gt_ci"][a][1])), int(r["num_spanning"][a])))
    assert vcf.vcf_record(loci[0], LocusResult(geno, [], [], [])) == TUTORIAL_RECORD


@pytest.mark.gpu
def test_example_bam_to_vcf_record_on_the_gpu():
    # BASELINE.json configs[0] end to end: example/ catalog + reference + BAM -> clipped reads -> trgt_locus_batch -> VCF record
    from trgt_amd import locus
    import pyvcf as vcf
    reads, loci, records = _example()
    res = locus.analyze_batch([reads.locus_inputs(l, records) for l in loci])
    assert [vcf.vcf_record(l, r) for l, r in zip(loci, res)] == [TUTORIAL_RECORD]
    # the cluster genotyper sees the same two alleles on this clean locus
    res = locus.analyze_batch([reads.locus_inputs(l, records) | {"genotyper": "cluster"} for l in loci])
    rec = vcf.vcf_record(loci[0], res[0]).split("\t")
    assert rec[4] == "C" + "CAG" * 11 and rec[9].startswith("1/1:33,33:")


@pytest.mark.gpu
def test_example_bam_through_the_native_ingestion_on_the_gpu():
    # the same end-to-end case with the native reader (trgt_amd/csrc/ingest.hip) in front: BAM + .bai, FASTA + .fai, catalog ->
    # trgt_ingest_batch arrays -> trgt_locus_batch -> VCF record of the tutorial
    from trgt_amd import ingest, locus
    import pyvcf as vcf
    reads, loci, _ = _example()
    b = ingest.Reader(os.path.join(EX, "sample.bam"), os.path.join(EX, "reference.fasta")).batch(os.path.join(EX, "repeat.bed"))
    out = locus.run_batch(b)
    assert [vcf.vcf_record(l, locus.locus_result(b, out, i)) for i, l in enumerate(loci)] == [TUTORIAL_RECORD]


@pytest.mark.gpu
def test_native_ingestion_with_4bit_reads_gives_the_same_call():
    # the reads of the BAM handed to the GPU as the 4-bit codes the records hold (keep_bam4 -> TRGT_READS_BAM4): same results, same record
    from trgt_amd import ingest, locus
    rd = ingest.Reader(os.path.join(EX, "sample.bam"), os.path.join(EX, "reference.fasta"))
    b = rd.batch(os.path.join(EX, "repeat.bed"), keep_bam4=1)
    ref = locus.run_batch(b)
    out = locus.run_batch(ingest.bam4_view(b))
    for f in ("span_start", "span_end", "n_alleles", "allele_len", "allele_blob", "ci", "num_spanning", "classification", "read_rank", "spans3", "motif_counts"):
        assert np.array_equal(getattr(out, f), getattr(ref, f)), f
    assert np.array_equal(out.purity.view(np.uint64), ref.purity.view(np.uint64))
    assert locus.locus_result(b, out, 0).vcf_fields() == locus.locus_result(b, ref, 0).vcf_fields()
    rd.close()


@pytest.mark.gpu
def test_native_writers_on_the_example(tmp_path):
    # catalog + reference + BAM -> native ingestion -> trgt_locus_batch -> native VCF and spanning-reads BAM (write_vcf.rs, write_bam.rs)
    import gzip
    from trgt_amd import ingest, locus, writers
    import pyreads
    rd = ingest.Reader(os.path.join(EX, "sample.bam"), os.path.join(EX, "reference.fasta"))
    b = rd.batch(os.path.join(EX, "repeat.bed"), keep_native=True)
    out = locus.run_batch(b)
    for vcf_name in ("out.vcf", "out.vcf.gz"):
        w = writers.Writer(rd, tmp_path / vcf_name, tmp_path / "out.spanning.bam", sample_name="sample", command_line="trgt genotype --test")
        w.write(b, out)
        w.close()
        text = (gzip.open if vcf_name.endswith(".gz") else open)(tmp_path / vcf_name, "rt").read().splitlines()
        assert text[0] == "##fileformat=VCFv4.2" and '##FORMAT=<ID=AM,Number=.,Type=Float,Description="Mean methylation level per allele">' in text
        assert "##contig=<ID=chrA,length=11061>" in text and "##trgtVersion=3.0.0" in text and "##trgtCommand=trgt genotype --test" in text
        assert text[-2] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample" and text[-1] == TUTORIAL_RECORD
    # the spanning reads: every kept read, clipped to 50 bases of flank on either side of its repeat, tagged as write_bam.rs:72-144 does
    recs = pyreads.read_bam(str(tmp_path / "out.spanning.bam"))
    res = locus.locus_result(b, out, 0)
    assert len(recs) == len(res.reads) == 29
    for k, (rec, ri) in enumerate(zip(recs, res.reads)):
        s, e = int(out.span_start[ri]), int(out.span_end[ri])
        full = bytes(b["read_blob"][int(b["read_off"][ri]):int(b["read_off"][ri]) + int(b["read_len"][ri])]).decode()
        assert rec.name == b["read_name"][ri] and rec.seq == full[s - 50:e + 50] and rec.contig == "chrA"
        assert sum(n for c, n in rec.cigar if c in pyreads.QRY_CONSUMING) == len(rec.seq)
        assert abs(rec.rq - b["read_qual"][ri]) < 1e-7
