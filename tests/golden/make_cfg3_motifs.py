#!/usr/bin/env python3
"""Build tests/golden/pathogenic_motif_sets.json: the motif sets (ID, MOTIFS, STRUC) of the reference's pathogenic repeat catalog,
repeats/pathogenic_repeats.hg38.bed:1-56 -- the catalog BASELINE.json configs[2] / SURVEY.md Appendix E "cfg3" draws its loci from.
Run in the build container only (needs /root/reference):  python tests/golden/make_cfg3_motifs.py"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
out = []
for line in open("/root/reference/repeats/pathogenic_repeats.hg38.bed"):
    contig, start, end, info = line.split()
    f = dict(x.split("=", 1) for x in info.split(";"))
    out.append(dict(id=f["ID"], motifs=f["MOTIFS"].split(","), struc=f["STRUC"], ref_len=int(end) - int(start)))
json.dump(dict(src="repeats/pathogenic_repeats.hg38.bed (ID, MOTIFS, STRUC, end - start)", loci=out),
          open(os.path.join(HERE, "pathogenic_motif_sets.json"), "w"), indent=1)
print(len(out), "motif sets")
