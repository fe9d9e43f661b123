"""BAM -> VCF through the three native stages on a synthetic data set (trgt_amd/synth_bam.py): ingestion -> trgt_locus_batch -> writer.
The alleles the reads were made from are known, so the calls can be checked without the oracle; the files must not depend on the writer's
thread count nor on whether the reads went up as ASCII or as 4-bit codes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bam_to_vcf_calls_the_alleles_the_reads_were_made_from(tmp_path):
    from trgt_amd import ingest, locus, synth_bam, writers
    ds = synth_bam.write_dataset(str(tmp_path / "ds"), n_loci=96, read_len=2500)
    rd = ingest.Reader(ds["bam"], ds["fasta"])
    b = rd.batch(ds["bed"], keep_native=True, keep_bam4=1, threads=4)
    out = locus.run_batch(b, locus.Params())
    out4 = locus.run_batch(ingest.bam4_view(b), locus.Params())
    for k in ("n_alleles", "allele_len", "span_start", "span_end", "classification", "ci", "allele_blob", "motif_counts", "spans3"):
        assert np.array_equal(getattr(out, k), getattr(out4, k)), k
    got = np.sort(out.allele_len.reshape(-1, 2).astype(np.int64), axis=1)
    assert (out.n_alleles == 2).all() and np.array_equal(got, np.sort(ds["allele_len"], axis=1))
    texts = []
    for th in (1, 6):
        w = writers.Writer(rd, tmp_path / ("o%d.vcf" % th), tmp_path / ("o%d.bam" % th), threads=th)
        w.write(b, out)
        w.close()
        texts.append((open(tmp_path / ("o%d.vcf" % th), "rb").read(), open(tmp_path / ("o%d.bam" % th), "rb").read()))
    assert texts[0] == texts[1]
    lines = [l for l in texts[0][0].decode().splitlines() if not l.startswith("#")]
    assert len(lines) == 96 and all(l.split("\t")[7].startswith("TRID=L%d;" % i) for i, l in enumerate(lines))
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from bamtools import read_bam_records
    _, _, recs = read_bam_records(str(tmp_path / "o1.bam"))
    assert len(recs) == 96 * 30 and all(r["tags"]["TR"][1].startswith("L") for r in recs[:10])
