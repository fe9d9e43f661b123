"""Host-side mirror of the VCF record the reference writes for one locus.

Reference: VcfWriter::write / add_locus_info / add_allele_info / add_missing_allele_info / set_gt and the encode_* helpers,
src/trgt/writers/write_vcf.rs:95-397.  The record is rendered as `bcftools view --no-header` prints it (tab-separated text), which
is how the reference documents its expected output (docs/tutorial.md:43-46).  AM (methylation) is "." per allele: methylation tags
are not carried through this path.
"""
from trgt_amd.hmm import encode_ap, encode_mc, encode_ms


def set_gt(locus_tr: bytes, genotype):
    """write_vcf.rs:219-260: allele sequences (reference first) and the GT indexes"""
    seqs, idx = [locus_tr], []
    for a in genotype:
        if a.seq == locus_tr:
            idx.append(0)
        elif len(seqs) == 1:
            idx.append(1)
            seqs.append(a.seq)
        elif genotype[0].seq == genotype[1].seq:
            idx.append(1)
        else:
            idx.append(2)
            seqs.append(a.seq)
    return seqs, idx


def vcf_record(locus, result, sample_meth=None):
    """locus: pyreads.Locus; result: trgt_amd.locus.LocusResult.  Returns the record as one tab-separated line."""
    pad = locus.left_flank[-1:]
    info = "TRID=%s;END=%d;MOTIFS=%s;STRUC=%s" % (locus.id, locus.end, ",".join(locus.motifs), locus.struc)
    fmt = "GT:AL:ALLR:SD:MC:MS:AP:AM"
    g = result.genotype
    if not g:  # add_missing_allele_info
        ref = (pad + locus.tr).decode()
        return "\t".join([locus.contig, str(locus.start), ".", ref, ".", ".", ".", info, fmt, ".:.:.:.:.:.:.:."])
    seqs, idx = set_gt(locus.tr, g)
    padded = [(pad + s).decode() for s in seqs]
    ann = [a.annotation for a in g]
    am = ",".join("." if a.meth is None else "%.2f" % a.meth for a in g)
    sample = ":".join(["/".join(str(i) for i in idx), ",".join(str(len(a.seq)) for a in g), ",".join("%d-%d" % a.ci for a in g),
                       ",".join(str(a.num_spanning) for a in g), encode_mc(ann), encode_ms(ann), encode_ap(ann), am])
    return "\t".join([locus.contig, str(locus.start), ".", padded[0], ",".join(padded[1:]) if len(padded) > 1 else ".", ".", ".", info,
                      fmt, sample])
