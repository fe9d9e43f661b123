#!/usr/bin/env python3
"""Generate tests/golden/read_kats.json: the known-answer tests the reference holds for the per-read steps either side of the hot path
(SURVEY.md 8(f) rows 3 / 4): clip_to_region (src/trgt/reads/clip_region.rs tests), clip_bases (clip_bases.rs tests), extract_snps_offset
(snp.rs tests, incl. the CIGAR of a read of the example data set), get_meth over MM / ML (read.rs tests), the CIGAR length helpers
(cigar.rs tests), utils::math::median (utils/math.rs tests) and GenomicRegion::from_string (utils/region.rs tests).

Run ONCE in the build container (needs /root/reference, absent on the GPU box).  The fixture is DATA: inputs and expected outputs pulled
out of the reference's test modules by regular expressions, each group tagged with the file:line it was read from; no source text is
stored."""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def rs(path):
    return open(os.path.join(REF, path)).read()


def ints(s):
    return [int(v) for v in re.findall(r"-?\d+", s)]


def test_bodies(src):
    tests = src[src.index("mod tests"):]
    for m in re.finditer(r"#\[test\]\s*fn (\w+)\(\) \{(.*?)\n    \}", tests, re.S):
        yield m.group(1), m.group(2), src[:src.index("fn " + m.group(1) + "(")].count("\n") + 1


READ = r'make_read\("(\w*)", (?:vec!\[([\d, ]*)\]|Vec::new\(\)), cigar\)'
CIG = r'let cigar = make_cigar\((\d+), "([^"]+)"\);'


def clip_cases(path, call_re, argnames):
    """Every `let cigar..; let read = make_read(..)` block followed by calls + expectations (None, or the `expected` read built just before)."""
    out = []
    for name, body, line in test_bodies(rs(path)):
        cur_read, pending_cigar, expected = None, None, None
        pos = 0
        toks = []
        for m in re.finditer(CIG + "|let (read|expected) = " + READ + r";|let expected = read\.clone\(\);|" + call_re, body):
            toks.append(m)
        for m in toks:
            t = m.group(0)
            if t.startswith("let cigar"):
                pending_cigar = (int(m.group(1)), m.group(2))
            elif t.startswith("let read") or (t.startswith("let expected") and "clone" not in t):
                rd = dict(bases=m.group(4), meth=ints(m.group(5) or ""), ref_pos=pending_cigar[0], cigar=pending_cigar[1])
                if t.startswith("let read"):
                    cur_read = rd
                else:
                    expected = rd
            elif "clone" in t:
                expected = dict(cur_read)
            else:  # a call: either inside assert_eq!(read.f(args), None|Some(expected)) or `let clipped_read = read.f(args)` + later assert
                args = ints(m.group("args"))
                tail = body[m.end():m.end() + 200]
                if re.match(r"\s*,\s*None\)", tail) or (t.startswith("let clipped") and re.search(r"assert_eq!\(clipped_read, None\)", tail)):
                    exp = None
                else:
                    if t.startswith("let clipped"):  # the expectation is built AFTER the call
                        em = re.search(CIG + r"\s*let expected = " + READ + ";", tail + body[m.end() + 200:])
                        if em and "read.clone" not in body[:m.start()].split("let read")[-1]:
                            expected = dict(bases=em.group(3), meth=ints(em.group(4) or ""), ref_pos=int(em.group(1)), cigar=em.group(2))
                    exp = dict(expected)
                out.append(dict(test=name, src="%s:%d" % (path, line), read=dict(cur_read), **dict(zip(argnames, args)), expected=exp))
    return out


def main():
    k = {}
    # ---- clip_to_region: clip_region.rs tests
    k["clip_to_region"] = clip_cases("src/trgt/reads/clip_region.rs",
                                     r"(?:let clipped_read = read|read)\.clip_to_region\(\((?P<args>\d+, \d+)\)\)", ("region_start", "region_end"))
    assert len(k["clip_to_region"]) == 7, len(k["clip_to_region"])
    # ---- clip_bases: clip_bases.rs tests
    k["clip_bases"] = clip_cases("src/trgt/reads/clip_bases.rs", r"read\.clip_bases\((?P<args>\d+, \d+)\)", ("left_len", "right_len"))
    assert len(k["clip_bases"]) == 15, len(k["clip_bases"])
    # ---- extract_snps_offset: snp.rs test_mismatch_count_offset_full_cigar (+ the op list of test_cigar_conversion == the CIGAR string)
    snp = rs("src/trgt/reads/snp.rs")
    cig = re.search(r'const CIGAR_STRING: &str = "([^"]+)";', snp).group(1)
    body = dict((n, (b, l)) for n, b, l in test_bodies(snp))
    conv = body["test_cigar_conversion"][0]
    ops = re.findall(r"CigarOp::(\w+)\((\d+)\)", conv)
    code = dict(Equal="=", Del="D", Ins="I", Diff="X", SoftClip="S", Match="M")
    assert "".join(n + code[o] for o, n in ops) == cig
    b, line = body["test_mismatch_count_offset_full_cigar"]
    k["extract_snps_offset"] = [dict(src="src/trgt/reads/snp.rs:%d" % line, ref_pos=int(re.search(r"ref_pos: (\d+)", b).group(1)), cigar=cig,
                                     region=[int(re.search(r"start: (\d+)", b).group(1)), int(re.search(r"end: (\d+)", b).group(1))],
                                     expected=ints(re.search(r"let ground_truth = vec!\[(.*?)\];", b, re.S).group(1)))]
    assert len(k["extract_snps_offset"][0]["expected"]) == 17
    # ---- get_meth: read.rs tests
    rd = rs("src/trgt/reads/read.rs")
    gm = []
    for name, b, line in test_bodies(rd):
        bases = re.search(r'let bases = b"(\w+)";', b).group(1)
        m = re.search(r'create_record\(bases, (?:"(\w+)"|mm), (?:&\[([\d, ]*)\]|&ml), (true|false)\)', b)
        mm = m.group(1) or re.search(r'let mm = "([^"]+)";', b).group(1)
        ml = ints(m.group(2)) if m.group(2) is not None else ints(re.search(r"let ml = \[([\d, ]*)\];", b).group(1))
        exp = None if "is_none()" in b else ints(re.search(r"Some\(vec!\[([\d, ]*)\]\)", b).group(1))
        gm.append(dict(test=name, src="src/trgt/reads/read.rs:%d" % line, bases=bases, mm=mm, ml=ml, reverse=m.group(3) == "true", expected=exp))
    assert [g["test"] for g in gm] == ["test_basemods_error", "test_matching_modifications"], gm
    k["get_meth"] = gm
    # ---- cigar.rs tests
    cg = rs("src/trgt/reads/cigar.rs")
    names = dict(Match="M", Ins="I", Del="D", SoftClip="S")
    lens = []
    for name, b, line in test_bodies(cg):
        if name == "test_query_len":
            ops = [(int(n), names[o]) for o, n in re.findall(r"CigarOp::(\w+)\((\d+)\)", b)]
            lens.append(dict(test=name, src="src/trgt/reads/cigar.rs:%d" % line, ops=ops, expected=int(re.search(r"query_len\(\), (\d+)", b).group(1))))
        else:
            fn = "get_ref_len" if "ref" in name else "get_query_len"
            for o, n, e in re.findall(r"CigarOp::(\w+)\((\d+)\)\.%s\(\), (\d+)" % fn, b):
                lens.append(dict(test=name, src="src/trgt/reads/cigar.rs:%d" % line, fn=fn, op=[int(n), names[o]], expected=int(e)))
    assert len(lens) == 9, len(lens)
    k["cigar_lens"] = lens
    # ---- utils/math.rs median tests (the literal cases; the randomised comparison with the naive median is a property, not a vector)
    mt = rs("src/utils/math.rs")
    med = []
    for name, b, line in test_bodies(mt):
        m = re.search(r"let data(?:: \[i32; 0\])? = \[([-\d, ]*)\];\s*assert_eq!\(median\(&data\), (None|Some\(([-\d.]+)\))\);", b)
        if m:
            med.append(dict(test=name, src="src/utils/math.rs:%d" % line, data=ints(m.group(1)), expected=None if m.group(2) == "None" else float(m.group(3))))
    assert len(med) >= 9, len(med)
    k["median"] = med
    # ---- utils/region.rs tests: what the catalog's coordinates must parse like
    rg = rs("src/utils/region.rs")
    reg = []
    for name, b, line in test_bodies(rg):
        m = re.search(r'GenomicRegion::from_string\("([^"]+)"\)', b)
        if not m:  # (GenomicRegion::new called directly: covered by the from_string case with the same interval)
            continue
        ok = "unwrap()" in b
        e = dict(test=name, src="src/utils/region.rs:%d" % line, encoding=m.group(1), ok=ok)
        if ok:
            e["contig"] = re.search(r'region\.contig, "(\w+)"', b).group(1)
            e["start"], e["end"] = int(re.search(r"region\.start, (\d+)", b).group(1)), int(re.search(r"region\.end, (\d+)", b).group(1))
        else:
            e["error"] = re.search(r'Err\("([^"]+)"', b).group(1)
        reg.append(e)
    assert len(reg) >= 3, reg
    k["region"] = reg
    with open(os.path.join(HERE, "read_kats.json"), "w") as f:
        json.dump(k, f, indent=1)
    print({a: len(b) for a, b in k.items()})


if __name__ == "__main__":
    main()
