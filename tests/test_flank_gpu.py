"""genotype_flank (src/trgt/genotype/genotype_flank.rs:9-290, applied by analyze at tr.rs:69-75) inside trgt_locus_batch: loci whose two
alleles are at most 10 bases apart are genotyped again from the reads' haplotype tags or from heterozygous SNVs of the flanks.  The
GPU path (device genotyper + host step) against the oracle's restatement, which the reference's two inline tests pin
(tests/test_oracle_caller_kats.py), and against those known answers directly."""
import json
import os

import numpy as np
import pytest

from helpers import mutate, rand_dna
from test_locus_gpu import _run_both
from test_oracle_caller_kats import decode_flank_read

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "caller_kats.json")))


def _oracle(oracle, L, params):
    meta = dict(hp_tag=L.get("hp_tag"), start_offset=L["start_offset"], end_offset=L["end_offset"], mismatch_offsets=L["mismatch_offsets"])
    return oracle.locus_analyze(L["left_flank"], L["right_flank"], L["tr"], L["motifs"], L["reads"], flank_len=params.search_flank_len,
                                min_flank_id_frac=params.min_flank_id_frac, max_depth=params.max_depth, scoring=params.aln_scoring,
                                ploidy=L.get("ploidy", 2), genotyper=1 if L.get("genotyper") == "cluster" else 0, meta=meta)


def _check(oracle, loci, expect_changed=None):
    from trgt_amd import locus
    params = locus.Params()
    b = locus.pack(loci)
    refs = [_oracle(oracle, L, params) for L in loci]
    plain = [oracle.locus_analyze(L["left_flank"], L["right_flank"], L["tr"], L["motifs"], L["reads"], ploidy=L.get("ploidy", 2),
                                  genotyper=1 if L.get("genotyper") == "cluster" else 0) for L in loci]
    changed = [r["alleles"] != q["alleles"] or list(r["classification"]) != list(q["classification"]) for r, q in zip(refs, plain)]
    if expect_changed is not None:
        assert changed == expect_changed
    for how, out in _run_both(locus, b, params):
        for l, ref in enumerate(refs):
            got = locus.locus_result(b, out, l)
            assert [a.seq.decode() for a in got.genotype] == ref["alleles"], (how, l)
            assert got.reads == [int(v) for v in ref["kept_read"]] and got.classification == [int(v) for v in ref["classification"]], (how, l)
            if ref["n_alleles"]:
                f = got.vcf_fields()
                for k in ("AL", "ALLR", "SD", "MC", "MS", "AP"):
                    assert f[k] == ref[k], (how, l, k)
    return refs, changed


def _kat_locus(rng, kat):
    lf, rf = rand_dna(rng, 250).replace(b"TA", b"GC"), rand_dna(rng, 250).replace(b"TA", b"GC")
    lf, rf = (lf + rand_dna(rng, 250))[:250], (rf + rand_dna(rng, 250))[:250]
    rd = [decode_flank_read(e) for e in kat["reads"]]
    return dict(left_flank=lf, right_flank=rf, tr=b"TATATATA", motifs=[b"TA"], ploidy=2,
                reads=[rand_dna(rng, 60) + lf + r[0] + rf + rand_dna(rng, 60) for r in rd],
                hp_tag=None, start_offset=[r[2] for r in rd], end_offset=[r[3] for r in rd], mismatch_offsets=[r[1] for r in rd])


def test_reference_kats_through_the_locus_path(oracle):
    rng = np.random.default_rng(4)
    loci = [_kat_locus(rng, k) for k in KATS["genotype_flank"]]
    refs, changed = _check(oracle, loci)
    exp = KATS["genotype_flank"][0]["expected"]
    # F1: the SNV split decides the genotype and the read assignment (reads in LocusResult order: sorted by repeat length, stable)
    assert refs[0]["alleles"] == exp["alleles"] and [tuple(c) for c in refs[0]["gt_ci"]] == [tuple(g["ci"]) for g in exp["gt"]]
    order = [int(v) for v in refs[0]["kept_read"]]
    assert [int(v) for v in refs[0]["classification"]] == [0, 0, 0, 1, 1, 1] and sorted(order[:3]) == [0, 1, 5]
    # F2: homozygous SNVs -> None: the length genotyper's call stands
    assert refs[1]["alleles"] == ["TATATATATA", "TATATATATA"] and not changed[1]


def _phased_locus(rng, motif, c1, c2, n=24, hp_frac=1.0, snv=False, err=0.004, genotyper="size"):
    lf, rf = rand_dna(rng, 250), rand_dna(rng, 250)
    alleles = [motif * c1, motif * c2]
    reads, hp, so, eo, mm = [], [], [], [], []
    for i in range(n):
        h = i % 2
        tr = mutate(rng, alleles[h], err, err / 2, err / 2)
        lc, rc = int(rng.integers(260, 700)), int(rng.integers(260, 700))
        reads.append(rand_dna(rng, lc - 250) + lf + tr + rf + rand_dna(rng, rc - 250))
        hp.append((h + 1) if rng.random() < hp_frac else None)
        so.append(-lc); eo.append(rc)
        offs = []
        if snv:
            if h == 1:
                offs += [-120, 85]       # a heterozygous SNV in either flank, on haplotype 2
            offs += [-200]               # a homozygous one
            if rng.random() < 0.1:
                offs += [int(rng.integers(90, 250))]  # a sequencing error
        mm.append(sorted(o for o in offs if -lc <= o <= rc))
    return dict(left_flank=lf, right_flank=rf, tr=alleles[0], motifs=[motif], ploidy=2, reads=reads, genotyper=genotyper,
                hp_tag=hp if hp_frac > 0 else None, start_offset=so, end_offset=eo, mismatch_offsets=mm)


def test_haplotype_tags_and_flank_snvs_decide_close_alleles(oracle):
    rng = np.random.default_rng(8)
    loci = [_phased_locus(rng, b"CAG", 20, 21),                         # tags split a 3-base difference
            _phased_locus(rng, b"CAG", 20, 20),                         # homozygous by length, tagged: two identical alleles either way
            _phased_locus(rng, b"AAG", 15, 17, hp_frac=0.5),            # too few tagged reads, no mismatches: None
            _phased_locus(rng, b"CAG", 20, 21, hp_frac=0.0, snv=True),  # no tags: the flank SNVs split the reads
            _phased_locus(rng, b"CAG", 12, 30, snv=True),               # alleles far apart: the step does not run
            _phased_locus(rng, b"CCG", 18, 19, hp_frac=0.8, snv=True, err=0.03),  # noisy repeats: no majority sequence -> repair_consensus
            _phased_locus(rng, b"AT", 25, 27, hp_frac=0.0, snv=True, genotyper="cluster"),
            _phased_locus(rng, b"CAG", 20, 21, n=9, hp_frac=0.75)]
    _check(oracle, loci)


def test_random_loci_with_read_metadata(oracle):
    rng = np.random.default_rng(21)
    loci = []
    for _ in range(40):
        m = rand_dna(rng, int(rng.integers(2, 7)))
        c1 = int(rng.integers(5, 40))
        c2 = max(3, c1 + int(rng.integers(-4, 5)))
        loci.append(_phased_locus(rng, m, c1, c2, n=int(rng.integers(6, 40)), hp_frac=float(rng.choice([0.0, 0.6, 0.9, 1.0])),
                                  snv=bool(rng.integers(0, 2)), err=float(rng.choice([0.002, 0.01, 0.03])),
                                  genotyper="cluster" if rng.random() < 0.2 else "size"))
    _check(oracle, loci)


def test_split_and_submitted_batches_keep_the_flank_genotype(oracle):
    """split_batch chunks and trgt_locus_batch_submit / _wait must apply genotype_flank like the blocking call on the whole
    batch (the per-read fields travel with the chunk / the ticket); chunks of a 4-bit batch keep their encoding."""
    from trgt_amd import locus
    from trgt_amd.driver import split_batch
    rng = np.random.default_rng(8)
    loci = [_phased_locus(rng, b"CAG", 20, 22, hp_frac=1.0), _phased_locus(rng, b"CAG", 20, 22, hp_frac=0.0, snv=True),
            _phased_locus(rng, b"GGC", 10, 12, hp_frac=0.9), _phased_locus(rng, b"AT", 25, 27, hp_frac=0.0, snv=True),
            _phased_locus(rng, b"CCG", 18, 19, hp_frac=0.8, snv=True, err=0.03)]
    b = locus.pack(loci)
    whole = locus.run_batch(b)
    recs = lambda bb, out: [(locus.locus_result(bb, out, l).genotype, locus.locus_result(bb, out, l).classification) for l in range(int(bb["n_loci"]))]
    want = recs(b, whole)
    plain = dict(b)
    for k in ("hp_tag", "start_offset", "end_offset", "mismatch_offsets", "mismatch_off", "_cin"):
        plain.pop(k, None)
    assert recs(plain, locus.run_batch(plain)) != want  # the metadata does change these genotypes
    got = []
    for c in split_batch(b, 2):
        got += recs(c, locus.run_batch(c))
    assert got == want
    assert recs(b, locus.submit_batch(b).wait()) == want
    pk = locus.pack_bam4(b)
    got4 = []
    for c in split_batch(pk, 2):
        got4 += recs(c, locus.run_batch(c))
    assert got4 == want
