#!/usr/bin/env python3
"""Generate tests/golden/caller_kats.json: the known-answer tests the reference holds for the callers of the hot path and for the
wrapper's accessors (G1-G4, S1, H12, W19).

Run ONCE in the build container (needs /root/reference, absent on the GPU box).  The fixture is DATA -- inputs and expected
outputs, each tagged with the reference file:line it was read from; the vectors are pulled out of the reference's test modules by
regular expressions rather than re-typed, and no reference source text is stored.
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def rs(path):
    return open(os.path.join(REF, path)).read()


def ints(s):
    return [int(v) for v in re.findall(r"-?\d+", s)]


def genotype_cases(path, ploidy):
    src = rs(path)
    tests = src[src.index("mod tests"):]
    out = []
    for m in re.finditer(r"fn (\w+)\(\) \{(.*?)\n    \}", tests, re.S):
        name, body = m.group(1), m.group(2)
        sizes = ints(re.search(r"let sizes = vec!\[(.*?)\]", body).group(1))
        counts = ints(re.search(r"let counts = vec!\[(.*?)\]", body).group(1))
        exp = [dict(size=int(a), ci=[int(b), int(c)]) for a, b, c in re.findall(r"TrSize::new\((\d+), \((\d+), (\d+)\)\)", body)]
        line = src[:src.index("fn " + name)].count("\n") + 1
        out.append(dict(id=name, src="%s:%d" % (path, line), ploidy=ploidy, sizes=sizes, counts=counts, expected=exp))
    return out


def main():
    kats = {}
    # G1-G3 haploid.rs:36-61, G4 diploid.rs:109-120
    kats["genotype"] = genotype_cases("src/trgt/genotype/haploid.rs", 1) + genotype_cases("src/trgt/genotype/diploid.rs", 2)
    assert [k["id"] for k in kats["genotype"]] == ["clean_tr", "mosaic_tr", "tr_with_outliers", "clean_het_tr"], kats["genotype"]
    # S1 span_locater.rs:72-130: (sequence, piece, expected span of the leftmost exact occurrence)
    sl = rs("src/trgt/genotype/span_locater.rs")
    block = sl[sl.index("let test_cases = vec!["):sl.index("for (seq_str, piece_str, expected_span)")]
    cases = []
    for m in re.finditer(r'\("([^"]*)", "([^"]*)", (None|Some\(\((\d+), (\d+)\)\))\)', block):
        cases.append(dict(seq=m.group(1), piece=m.group(2), span=None if m.group(3) == "None" else [int(m.group(4)), int(m.group(5))]))
    assert len(cases) == 13, len(cases)
    kats["exact_search"] = dict(src="src/trgt/genotype/span_locater.rs:%d" % (sl[:sl.index("let test_cases")].count("\n") + 1), cases=cases)
    # H10-H12 events.rs:124-145: get_base_match
    ev = rs("src/hmm/events.rs")
    h = []
    for name, motif, state, exp in re.findall(r'fn (\w+)\(\) \{\s*let motifs = vec!\["(\w)"\.as_bytes\(\)\.to_vec\(\)\];\s*let hmm = build_hmm\(&motifs\);\s*'
                                              r"assert_eq!\(get_base_match\(&hmm, (\d+)\), b'(.)'\);", ev):
        h.append(dict(id=name, motifs=[motif], state=int(state), expected=exp))
    m = re.search(r"fn (silent_states_match_a_blank_character)\(\) \{(.*?)\n    \}", ev, re.S)
    body = m.group(2)
    n_states = int(re.search(r"Hmm::new\((\d+)\)", body).group(1))
    ems = [[0.0] * 5 for _ in range(n_states)]
    for st, vals in re.findall(r"set_ems\((\d+), vec!\[(.*?)\]\)", body):
        ems[int(st)] = [float(v) for v in vals.split(",")]
    st, exp = re.search(r"get_base_match\(&hmm, (\d+)\), b'(.)'", body).groups()
    h.append(dict(id=m.group(1), n_states=n_states, ems=ems, state=int(st), expected=exp))
    assert len(h) == 3, h
    kats["base_match"] = dict(src="src/hmm/events.rs:%d" % (ev[:ev.index("fn states_match_the_most_likely_base")].count("\n") + 1), cases=h)
    # W19 wfaligner.rs:1423-1435, 1456-1587: accessors of the wrapper (builder -> get_penalties; set_heuristic accepts every variant)
    wf = rs("src/wfaligner.rs")
    pen = []
    for t in ("fn test_get_penalties", "fn test_builder_pattern"):
        seg = wf[wf.index(t):]
        seg = seg[:seg.index("\n    }\n") + 1]
        for m in re.finditer(r"\.(edit|indel|linear|linear_with_match|affine|affine_with_match|affine2p|affine2p_with_match)\(([-\d, ]*)\)"
                             r"(?:\s*\.with_heuristic\(Heuristic::(\w+)\(([\d, ]*)\)\))?\s*\.build\(\);\s*assert_eq!\(\s*\w+\.get_penalties\(\),\s*"
                             r"Penalties::(\w+)(?:\s*\{(.*?)\})?\s*\);", seg, re.S):
            fields = {k: int(v) for k, v in re.findall(r"(\w+): (-?\d+)", m.group(6) or "")}
            pen.append(dict(builder=m.group(1), args=ints(m.group(2)), heuristic=[m.group(3)] + ints(m.group(4) or "") if m.group(3) else None,
                            kind=m.group(5), fields=fields))
    assert len(pen) == 11, len(pen)
    seg = wf[wf.index("fn test_set_heuristic"):]
    seg = seg[:seg.index("\n    }\n")]
    heur = [dict(kind=k, args=ints(a)) for k, a in re.findall(r"set_heuristic\(Heuristic::(\w+)(?:\(([\d, ]*)\))?\)", seg)]
    assert len(heur) == 7, heur
    kats["wrapper_accessors"] = dict(src="src/wfaligner.rs:%d" % (wf[:wf.index("fn test_set_heuristic")].count("\n") + 1), penalties=pen,
                                     set_heuristic=heur)
    # F1 / F2 genotype_flank.rs:343-390: reads given as encodings ('=' flank base, 'X' mismatching flank base, ACGT = the repeat; a
    # read is the repeat bases, its mismatch offsets are the X positions relative to the start (left) / end (right) of the repeat,
    # start_offset = -(bases before the repeat), end_offset = bases after it, no HP tag), expected Some((gt, alleles, assignment)) / None
    gf = rs("src/trgt/genotype/genotype_flank.rs")
    tests = gf[gf.index("mod tests"):]
    flank = []
    for m in re.finditer(r"fn (if_\w+)\(\) \{(.*?)\n    \}", tests, re.S):
        name, body = m.group(1), m.group(2)
        enc = re.findall(r'"([=XACGT]+)"', body[:body.index("let tr_seqs")])
        exp = None
        if "assert_eq!(result, None)" not in body:
            gt = [dict(size=int(a), ci=[int(b), int(c)]) for a, b, c in re.findall(r"size: (\d+),\s*ci: \((\d+), (\d+)\)", body)]
            alleles = re.findall(r'"([ACGT]+)"\.to_string\(\)', body)
            assignment = ints(re.search(r"let assignment = vec!\[(.*?)\]", body).group(1))
            exp = dict(gt=gt, alleles=alleles, assignment=assignment)
        flank.append(dict(id=name, src="src/trgt/genotype/genotype_flank.rs:%d" % (gf[:gf.index("fn " + name)].count("\n") + 1), reads=enc, expected=exp))
    assert [f["id"] for f in flank] == ["if_het_snvs_then_genotype", "if_hom_snvs_then_none"] and len(flank[0]["reads"]) == 6 and len(flank[1]["reads"]) == 4, flank
    kats["genotype_flank"] = flank
    json.dump(kats, open(os.path.join(HERE, "caller_kats.json"), "w"), indent=1)
    print({k: (len(v) if isinstance(v, list) else {a: (len(b) if isinstance(b, list) else b) for a, b in v.items()}) for k, v in kats.items()})


if __name__ == "__main__":
    main()
