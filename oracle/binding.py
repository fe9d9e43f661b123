"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package trgt_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

METRICS = {"indel": 0, "edit": 1, "linear": 2, "affine": 3, "affine2p": 4}
MEMORY = {"high": 0, "med": 1, "low": 2, "ultralow": 3}


class WfaParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "metric", "mismatch", "gap_open1", "gap_ext1", "gap_open2", "gap_ext2", "span", "pattern_begin_free",
        "pattern_end_free", "text_begin_free", "text_end_free", "scope", "memory_mode", "heuristic",
        "h_min_wavefront_length", "h_max_distance_threshold", "h_steps_between_cutoffs", "bialign_min_score",
        "bialign_min_length")]


class LocusParams(C.Structure):
    _fields_ = [("flank_len", C.c_int32), ("min_flank_id_frac", C.c_double), ("max_depth", C.c_int32),
                ("mism", C.c_int32), ("gapo", C.c_int32), ("gape", C.c_int32), ("ploidy", C.c_int32),
                ("genotyper", C.c_int32), ("min_read_qual", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("hmm.cpp", "wfa.cpp", "locus.cpp", "oracle.h", "oracle_internal.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_hmm_purity.restype = C.c_double
    return _LIB


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8) if len(b) else np.zeros(0, np.uint8)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def motif_blob(motifs):
    motifs = [m.encode() if isinstance(m, str) else bytes(m) for m in motifs]
    off = np.zeros(len(motifs) + 1, np.uint32)
    off[1:] = np.cumsum([len(m) for m in motifs])
    return _u8(b"".join(motifs)), off


def wfa_params(metric="affine", x=0, o1=0, e1=0, o2=0, e2=0, span="end2end", pbf=0, pef=0, tbf=0, tef=0,
               scope="alignment", memory="high", heuristic="default", min_score=250, min_length=100):
    p = WfaParams()
    lib().orc_wfa_default_params(C.byref(p))
    p.metric = METRICS[metric]
    if metric in ("linear", "affine", "affine2p"):
        p.mismatch, p.gap_open1, p.gap_ext1, p.gap_open2, p.gap_ext2 = x, o1, e1, o2, e2
    p.span = 1 if span == "endsfree" else 0
    p.pattern_begin_free, p.pattern_end_free, p.text_begin_free, p.text_end_free = pbf, pef, tbf, tef
    p.scope = 1 if scope == "alignment" else 0
    p.memory_mode = MEMORY[memory]
    if heuristic == "none":
        p.heuristic = 0
    elif heuristic != "default":
        p.heuristic = 1
        p.h_min_wavefront_length, p.h_max_distance_threshold, p.h_steps_between_cutoffs = heuristic
    p.bialign_min_score, p.bialign_min_length = min_score, min_length
    return p


def wfa_align(p, pattern, text):
    pa, te = _u8(pattern), _u8(text)
    ops = np.zeros(len(pa) + len(te) + 1, np.uint8)
    score, ops_len, n_match = C.c_int32(), C.c_int32(), C.c_int32()
    span4 = np.zeros(4, np.uint32)
    cells = C.c_int64()
    st = lib().orc_wfa_align(C.byref(p), _p(pa), len(pa), _p(te), len(te), C.byref(score), _p(ops), C.byref(ops_len),
                             C.byref(n_match), _p(span4), C.byref(cells))
    return dict(status=st, score=score.value, ops=bytes(ops[:ops_len.value]).decode(), n_match=n_match.value,
                span=[int(v) for v in span4], cells=cells.value)


def cigar_string(ops):
    out, i = [], 0
    while i < len(ops):
        j = i
        while j < len(ops) and ops[j] == ops[i]:
            j += 1
        out.append("%d%s" % (j - i, ops[i]))
        i = j
    return "".join(out)


def cigar_rle(ops, show_mismatches=True):
    o = _u8(ops.encode())
    out = np.zeros(len(o) + 1, np.uint32)
    n = lib().orc_cigar_rle(_p(o), len(o), int(show_mismatches), _p(out), len(out))
    return [int(v) for v in out[:n]]


def cigar_score(p, ops, clipped=None):
    o = _u8(ops.encode())
    if clipped is None:
        return lib().orc_cigar_score(C.byref(p), _p(o), len(o))
    return lib().orc_cigar_score_clipped(C.byref(p), _p(o), len(o), int(clipped))


# ------------------------------------------------------------------ HMM
def hmm_label(motifs, seq):
    mb, mo = motif_blob(motifs)
    s = _u8(seq.encode())
    cap = 64 + (len(s) + 2) * (max(len(m) for m in motifs) + 6)
    path = np.zeros(cap, np.int32)
    n = lib().orc_hmm_label(_p(mb), _p(mo), len(motifs), _p(s), len(s), _p(path), cap)
    assert n >= 0
    return path[:n].copy()


def hmm_remove_imperfect(motifs, path, seq, max_motif_len=6):
    mb, mo = motif_blob(motifs)
    s = _u8(seq.encode())
    path = np.ascontiguousarray(path, np.int32)
    out = np.zeros(len(path) * 3 + 16, np.int32)
    n = lib().orc_hmm_remove_imperfect(_p(mb), _p(mo), len(motifs), _p(path), len(path), _p(s), len(s), max_motif_len,
                                       _p(out), len(out))
    assert n >= 0
    return out[:n].copy()


def hmm_label_motifs(motifs, path):
    mb, mo = motif_blob(motifs)
    path = np.ascontiguousarray(path, np.int32)
    out = np.zeros(3 * (len(path) + 1), np.int32)
    n = lib().orc_hmm_label_motifs(_p(mb), _p(mo), len(motifs), _p(path), len(path), _p(out), len(path) + 1)
    assert n >= 0
    return out[:3 * n].reshape(-1, 3).copy()


def hmm_purity(motifs, path, seq):
    mb, mo = motif_blob(motifs)
    s = _u8(seq.encode())
    path = np.ascontiguousarray(path, np.int32)
    e, m = C.c_int32(), C.c_int32()
    p = lib().orc_hmm_purity(_p(mb), _p(mo), len(motifs), _p(path), len(path), _p(s), len(s), C.byref(e), C.byref(m))
    return p, e.value, m.value


def hmm_events(motifs, path, seq):
    mb, mo = motif_blob(motifs)
    s = _u8(seq.encode())
    path = np.ascontiguousarray(path, np.int32)
    out = np.zeros(len(path) * 4 + 64, np.uint8)
    n = lib().orc_hmm_events(_p(mb), _p(mo), len(motifs), _p(path), len(path), _p(s), len(s), _p(out), len(out))
    assert n >= 0
    return out[:n].copy()


def hmm_base_match(motifs, state):
    mb, mo = motif_blob(motifs)
    return chr(lib().orc_hmm_base_match(_p(mb), _p(mo), len(motifs), state))


def hmm_annotate(motifs, seq):
    """label_with_hmm for one allele (tr.rs:463-488)."""
    mb, mo = motif_blob(motifs)
    s = _u8(seq.encode() if isinstance(seq, str) else seq)
    cap = 64 + (len(s) + 2) * (max(len(m) for m in motifs) + 6)
    path = np.zeros(cap, np.int32)
    spans = np.zeros(3 * (len(s) + 2), np.int32)
    counts = np.zeros(len(motifs), np.int32)
    pl, ns, e, m = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    pur, cells = C.c_double(), C.c_int64()
    rc = lib().orc_hmm_annotate(_p(mb), _p(mo), len(motifs), _p(s), len(s), _p(path), cap, C.byref(pl), _p(spans),
                                len(s) + 2, C.byref(ns), _p(counts), C.byref(pur), C.byref(e), C.byref(m), C.byref(cells))
    assert rc == 0
    return dict(path=path[:pl.value].copy(), spans=spans[:3 * ns.value].reshape(-1, 3).copy(), counts=counts,
                purity=pur.value, edit=e.value, maxd=m.value, cells=cells.value)


def hmm_batch(batch, n_threads=1, want_path=True):
    """batch: dict with the product-ABI arrays (see trgt_amd.hmm.pack_hmm_batch)."""
    n = len(batch["job_set"])
    path = np.zeros(int(batch["path_off"][-1]) if want_path else 0, np.uint16)
    path_len = np.zeros(n, np.uint32)
    spans = np.zeros(3 * int(batch["span_off"][-1]), np.int32)
    n_spans = np.zeros(n, np.uint32)
    counts = np.zeros(int(batch["count_off"][-1]), np.uint32)
    purity = np.zeros(n, np.float64)
    edit = np.zeros(n, np.int32)
    maxd = np.zeros(n, np.int32)
    cells = C.c_int64()
    lib().orc_hmm_batch(len(batch["set_motif_begin"]) - 1, _p(batch["motif_blob"]), _p(batch["motif_off"]),
                        _p(batch["set_motif_begin"]), C.c_int64(n), _p(batch["job_set"]), _p(batch["seq_blob"]),
                        _p(batch["seq_off"]), _p(batch["seq_len"]), _p(path) if want_path else None, _p(batch["path_off"]),
                        _p(path_len), _p(spans), _p(batch["span_off"]), _p(n_spans), _p(counts), _p(batch["count_off"]),
                        _p(purity), _p(edit), _p(maxd), C.byref(cells), n_threads)
    return dict(path=path, path_len=path_len, spans=spans, n_spans=n_spans, counts=counts, purity=purity, edit=edit,
                maxd=maxd, cells=cells.value)


def wfa_batch(p, batch, n_threads=1, want_ops=True):
    """batch: dict(seqs, pat_off, pat_len, txt_off, txt_len, cigar_off, ops_off) -- product ABI layout."""
    n = len(batch["pat_len"])
    status = np.zeros(n, np.int32)
    score = np.zeros(n, np.int32)
    n_match = np.zeros(n, np.int32)
    span4 = np.zeros(4 * n, np.uint32)
    cigar = np.zeros(int(batch["cigar_off"][-1]), np.uint32)
    cigar_len = np.zeros(n, np.uint32)
    ops = np.zeros(int(batch["ops_off"][-1]) if want_ops else 0, np.uint8)
    ops_len = np.zeros(n, np.uint32)
    cells = C.c_int64()
    lib().orc_wfa_batch(C.byref(p), C.c_int64(n), _p(batch["seqs"]), _p(batch["pat_off"]), _p(batch["pat_len"]),
                        _p(batch["txt_off"]), _p(batch["txt_len"]), _p(status), _p(score), _p(n_match), _p(span4), _p(cigar),
                        _p(batch["cigar_off"]), _p(cigar_len), _p(ops) if want_ops else None, _p(batch["ops_off"]),
                        _p(ops_len), C.byref(cells), n_threads)
    return dict(status=status, score=score, n_match=n_match, span4=span4.reshape(-1, 4), cigar=cigar, cigar_len=cigar_len,
                ops=ops, ops_len=ops_len, cells=cells.value)


def find_spans(piece, reads, mism=2, gapo=5, gape=1, threshold=175.0):
    blob = _u8(b"".join(reads))
    lens = np.array([len(r) for r in reads], np.uint32)
    off = np.zeros(len(reads), np.uint64)
    off[1:] = np.cumsum(lens[:-1])
    pc = _u8(piece)
    st = np.zeros(len(reads), np.int32)
    en = np.zeros(len(reads), np.int32)
    used = np.zeros(len(reads), np.int32)
    cells = C.c_int64()
    lib().orc_find_spans(_p(pc), len(pc), C.c_int64(len(reads)), _p(blob), _p(off), _p(lens), mism, gapo, gape,
                         C.c_double(threshold), _p(st), _p(en), _p(used), C.byref(cells))
    return st, en, used, cells.value


def locus_analyze(left_flank, right_flank, ref_tr, motifs, reads, flank_len=250, min_flank_id_frac=0.7, max_depth=250,
                  scoring=(2, 5, 1), ploidy=2, genotyper=0, min_read_qual=0.98, read_qual=None, meta=None):
    """meta: dict(hp_tag, start_offset, end_offset, mismatch_offsets (list of lists)) per input read -> genotype_flank runs (tr.rs:69-75)"""
    p = LocusParams(flank_len, min_flank_id_frac, max_depth, scoring[0], scoring[1], scoring[2], ploidy, genotyper, min_read_qual)
    rq = None if read_qual is None else np.ascontiguousarray(read_qual, np.float64)
    mb, mo = motif_blob(motifs)
    blob = _u8(b"".join(reads))
    lens = np.array([len(r) for r in reads], np.uint32)
    off = np.zeros(len(reads), np.uint64)
    if len(reads) > 1:
        off[1:] = np.cumsum(lens[:-1])
    lf, rf, tr = _u8(left_flank), _u8(right_flank), _u8(ref_tr)
    n = len(reads)
    ss, se = np.zeros(n, np.int32), np.zeros(n, np.int32)
    cap = int(lens.max()) + 8 if n else 8
    a0, a1 = C.create_string_buffer(cap), C.create_string_buffer(cap)
    gt_size, gt_ci, by_hap = np.zeros(2, np.int32), np.zeros(4, np.int32), np.zeros(2, np.int32)
    kept, cls = np.zeros(n, np.int32), np.zeros(n, np.int32)
    n_alleles, n_sp = C.c_int32(), C.c_int32()
    scap = 64 * 1024
    mc, ms, ap = C.create_string_buffer(scap), C.create_string_buffer(scap), C.create_string_buffer(scap)
    stats = np.zeros(8, np.int64)
    keep = []
    cm = _read_meta(meta.get("hp_tag"), meta["start_offset"], meta["end_offset"], meta["mismatch_offsets"], keep) if meta is not None else None
    rc = lib().orc_locus_analyze_meta(C.byref(p), _p(lf), len(lf), _p(rf), len(rf), _p(tr), len(tr), _p(mb), _p(mo), len(motifs),
                                      C.c_int64(n), _p(blob), _p(off), _p(lens), _p(ss), _p(se), C.byref(n_alleles), a0, a1, cap,
                                      _p(gt_size), _p(gt_ci), C.byref(n_sp), _p(kept), _p(cls), _p(by_hap), mc, ms, ap, scap,
                                      _p(stats), _p(rq) if rq is not None else None, C.byref(cm) if cm is not None else None)
    assert rc == 0, rc
    na, k = n_alleles.value, n_sp.value
    alleles = [a0.value.decode(), a1.value.decode()][:na]
    return dict(span_start=ss, span_end=se, n_alleles=na, alleles=alleles, gt_size=gt_size[:na].copy(),
                gt_ci=gt_ci[:2 * na].reshape(-1, 2).copy(), kept_read=kept[:k].copy(), classification=cls[:k].copy(),
                num_spanning=by_hap[:na].copy(), MC=mc.value.decode(), MS=ms.value.decode(), AP=ap.value.decode(),
                AL=",".join(str(len(a)) for a in alleles),
                ALLR=",".join("%d-%d" % (gt_ci[2 * i], gt_ci[2 * i + 1]) for i in range(na)),
                SD=",".join(str(int(v)) for v in by_hap[:na]),
                stats=dict(wfa_cells=int(stats[0]), viterbi_cells=int(stats[1]), n_wfa_flank=int(stats[2]),
                           n_wfa_cons=int(stats[3]), bytes_io=int(stats[4]), n_wfa_ed=int(stats[5]),
                           n_purity=int(stats[6])))


class ReadMeta(C.Structure):
    _fields_ = [("hp_tag", C.c_void_p), ("start_offset", C.c_void_p), ("end_offset", C.c_void_p), ("mismatch_offsets", C.c_void_p),
                ("mismatch_off", C.c_void_p)]


def _read_meta(hp_tag, start_offset, end_offset, mismatch_lists, keep):
    """orc_read_meta from per-read Python values; `keep` collects the arrays that must outlive the call"""
    n = len(mismatch_lists)
    hp = np.array([-1 if v is None else v for v in (hp_tag if hp_tag is not None else [None] * n)], np.int16)
    so, eo = np.ascontiguousarray(start_offset, np.int32), np.ascontiguousarray(end_offset, np.int32)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum([len(m) for m in mismatch_lists])
    mm = np.array([v for m in mismatch_lists for v in m] + [0], np.int32)
    keep += [hp, so, eo, off, mm]
    return ReadMeta(hp.ctypes.data, so.ctypes.data, eo.ctypes.data, mm.ctypes.data, off.ctypes.data)


def genotype_flank(trs, hp_tag, start_offset, end_offset, mismatch_lists):
    """genotype_flank::genotype (genotype_flank.rs:9-42) -> None or dict(gt=[(size, (lo, hi))], alleles, assignment)"""
    n = len(trs)
    keep = []
    meta = _read_meta(hp_tag, start_offset, end_offset, mismatch_lists, keep)
    blob = _u8(b"".join(trs))
    lens = np.array([len(t) for t in trs], np.uint32)
    off = np.zeros(max(n, 1), np.uint64)
    if n > 1:
        off[1:] = np.cumsum(lens[:-1])
    cap = int(sum(len(t) for t in trs)) + 64
    a0, a1 = C.create_string_buffer(cap), C.create_string_buffer(cap)
    sizes, ci, asg = np.zeros(2, np.int32), np.zeros(4, np.int32), np.zeros(max(n, 1), np.int32)
    rc = lib().orc_genotype_flank(n, _p(blob), _p(off), _p(lens), C.byref(meta), _p(sizes), _p(ci), a0, a1, cap, _p(asg))
    assert rc >= 0, rc
    if rc == 0:
        return None
    return dict(gt=[(int(sizes[a]), (int(ci[2 * a]), int(ci[2 * a + 1]))) for a in range(2)], alleles=[a0.value.decode(), a1.value.decode()],
                assignment=[int(v) for v in asg[:n]])


def locus_analyze_many(batch, first, n, threads, flank_len=250, min_flank_id_frac=0.7, max_depth=250, scoring=(2, 5, 1)):
    """analyze_tr for loci [first, first + n) of a packed batch (trgt_amd.locus.pack / synth.generate layout) on `threads` native
    threads.  Returns (loci analysed, alleles called)."""
    p = LocusParams(flank_len, min_flank_id_frac, max_depth, scoring[0], scoring[1], scoring[2], 2, 0, 0.98)
    al = C.c_int64()
    f = lib().orc_locus_analyze_many
    f.restype = C.c_int64
    done = f(C.byref(p), C.c_int64(first), C.c_int64(n), _p(batch["flank_blob"]), _p(batch["lf_off"]), _p(batch["lf_len"]), _p(batch["rf_off"]),
             _p(batch["rf_len"]), _p(batch["tr_blob"]), _p(batch["tr_off"]), _p(batch["tr_len"]), _p(batch["motif_blob"]), _p(batch["motif_off"]),
             _p(batch["set_motif_begin"]), _p(batch["locus_read_begin"]), _p(batch["read_blob"]), _p(batch["read_off"]), _p(batch["read_len"]),
             int(threads), C.byref(al))
    return int(done), int(al.value)


def locus_records(batch, first, n, threads, stride=16384, flank_len=250, min_flank_id_frac=0.7, max_depth=250, scoring=(2, 5, 1),
                  min_read_qual=0.98):
    """analyze_tr for loci [first, first + n) of a packed batch; one text record per locus (see oracle.h) for whole-catalog
    comparisons against trgt_locus_batch (tests/tools/parity_sweep.py, trgt_amd-independent)."""
    p = LocusParams(flank_len, min_flank_id_frac, max_depth, scoring[0], scoring[1], scoring[2], 2, 0, min_read_qual)
    blob = C.create_string_buffer(int(n) * int(stride))
    f = lib().orc_locus_analyze_records
    f.restype = C.c_int64
    gt = batch.get("genotyper")
    rq = batch.get("read_qual")
    rq = None if rq is None else np.ascontiguousarray(rq, np.float64)
    done = f(C.byref(p), C.c_int64(first), C.c_int64(n), _p(batch["flank_blob"]), _p(batch["lf_off"]), _p(batch["lf_len"]), _p(batch["rf_off"]),
             _p(batch["rf_len"]), _p(batch["tr_blob"]), _p(batch["tr_off"]), _p(batch["tr_len"]), _p(batch["motif_blob"]), _p(batch["motif_off"]),
             _p(batch["set_motif_begin"]), _p(batch["locus_read_begin"]), _p(batch["read_blob"]), _p(batch["read_off"]), _p(batch["read_len"]),
             int(threads), _p(gt) if gt is not None else None, _p(batch["ploidy"]), blob, C.c_uint64(stride),
             _p(rq) if rq is not None else None)
    assert done == n, (done, n)
    raw = blob.raw
    return [raw[i * stride:raw.index(b"\0", i * stride)].decode() for i in range(int(n))]


def ward_linkage(dists, n):
    """kodama-style linkage(.., Method::Ward) on a condensed matrix.  Returns (steps[n-1,3] = cluster1, cluster2, size;
    dissimilarity[n-1]; the matrix as the call leaves it)."""
    d = np.ascontiguousarray(dists, np.float64).copy()
    st = np.zeros((max(n - 1, 1), 3), np.int32)
    di = np.zeros(max(n - 1, 1), np.float64)
    k = lib().orc_ward_linkage(_p(d), int(n), _p(st), _p(di))
    return st[:k], di[:k], d


def cluster_groups(dists, n):
    d = np.ascontiguousarray(dists, np.float64).copy()
    g = np.zeros(n, np.int32)
    k = lib().orc_cluster_groups(_p(d), int(n), _p(g))
    return k, g, d


def genotype_sizes(ploidy, sizes, counts):
    """haploid::genotype / diploid::genotype on a length histogram -> [(size, (ci_lo, ci_hi)), ...]"""
    s, c = np.ascontiguousarray(sizes, np.int32), np.ascontiguousarray(counts, np.int32)
    gt = np.zeros(6, np.int32)
    n = lib().orc_genotype_sizes(int(ploidy), _p(s), _p(c), len(s), _p(gt))
    return [(int(gt[3 * a]), (int(gt[3 * a + 1]), int(gt[3 * a + 2]))) for a in range(n)]


def hmm_base_match_ems(ems_probs, state):
    """get_base_match on Hmm::new(len(ems_probs)) with set_ems(state, probs) for every state; returns a 1-char string."""
    e = np.ascontiguousarray(ems_probs, np.float64).reshape(-1, 5)
    return chr(lib().orc_hmm_base_match_ems(len(e), _p(e), int(state)))
