"""Per-read helpers of the ingestion / writer steps on their own (include/trgt_hip.h, "per-read helpers"): ctypes bindings named after
the reference functions they replace (PacificBiosciences/trgt v3.0.0):
  CigarOpExt::get_ref_len / get_query_len, Cigar::query_len   src/trgt/reads/cigar.rs:8-45
  extract_snps_offset                                          src/trgt/reads/snp.rs:51-79
  get_meth                                                     src/trgt/reads/read.rs:55-96
  HiFiRead::clip_to_region                                     src/trgt/reads/clip_region.rs:19-184
  HiFiRead::clip_bases                                         src/trgt/reads/clip_bases.rs:9-120
  utils::math::median                                          src/utils/math.rs:73-98
CIGARs are lists of (length, op char) here and BAM words (len << 4 | code) at the ABI."""
import ctypes as C
import re

import numpy as np

from . import _lib

_OPS = "MIDNSHP=X"


def cigar_words(text_or_ops):
    """'3=2D2=' or [(3, '='), ...] -> uint32 BAM words."""
    ops = [(int(n), c) for n, c in re.findall(r"(\d+)([MIDNSHP=X])", text_or_ops)] if isinstance(text_or_ops, str) else list(text_or_ops)
    return np.array([(n << 4) | _OPS.index(c) for n, c in ops], np.uint32)


def cigar_text(words):
    return "".join("%d%s" % (int(w) >> 4, _OPS[int(w) & 15]) for w in words)


def _L():
    L = _lib.lib()
    if not getattr(L, "_readops_ready", False):
        VP, I64 = C.c_void_p, C.c_int64
        L.trgt_cigar_ref_len.argtypes = [C.c_uint32]; L.trgt_cigar_ref_len.restype = I64
        L.trgt_cigar_query_len.argtypes = [C.c_uint32]; L.trgt_cigar_query_len.restype = I64
        L.trgt_cigar_total_query_len.argtypes = [VP, I64]; L.trgt_cigar_total_query_len.restype = I64
        L.trgt_read_mismatch_offsets.argtypes = [VP, I64, I64, I64, I64, VP, I64]; L.trgt_read_mismatch_offsets.restype = I64
        L.trgt_read_meth.argtypes = [VP, I64, C.c_char_p, VP, I64, C.c_int32, VP, I64]; L.trgt_read_meth.restype = I64
        for f in (L.trgt_read_clip_to_region, L.trgt_read_clip_bases):
            f.argtypes = [VP, VP, I64, VP, I64, VP, I64, I64, I64, I64, VP, VP, VP, C.POINTER(I64), VP, C.POINTER(I64), C.POINTER(I64)]
            f.restype = I64
        L.trgt_median_i32.argtypes = [VP, I64, C.POINTER(C.c_float)]; L.trgt_median_i32.restype = C.c_int32
        L._readops_ready = True
    return L


def get_ref_len(op):
    return int(_L().trgt_cigar_ref_len(int(cigar_words([op])[0])))


def get_query_len(op):
    return int(_L().trgt_cigar_query_len(int(cigar_words([op])[0])))


def query_len(cigar):
    w = cigar_words(cigar)
    return int(_L().trgt_cigar_total_query_len(_lib.ptr(w) if len(w) else None, len(w)))


def extract_snps_offset(cigar, ref_pos, region_start, region_end):
    w = cigar_words(cigar)
    cap = int(sum(int(x) >> 4 for x in w)) + 1
    out = np.zeros(cap, np.int32)
    n = _L().trgt_read_mismatch_offsets(_lib.ptr(w) if len(w) else None, len(w), ref_pos, region_start, region_end, _lib.ptr(out), cap)
    if n < 0:
        raise _lib.TrgtHipError("trgt_read_mismatch_offsets: %d" % n)
    return [int(v) for v in out[:n]]


def get_meth(bases, mm, ml, reverse=False):
    b = np.frombuffer(bytes(bases), np.uint8)
    m = np.array(list(ml), np.uint8)
    out = np.zeros(len(b) + 1, np.uint8)
    n = _L().trgt_read_meth(_lib.ptr(b) if len(b) else None, len(b), mm.encode() if isinstance(mm, str) else mm, _lib.ptr(m) if len(m) else None, len(m),
                            1 if reverse else 0, _lib.ptr(out), len(out))
    if n < -1:
        raise _lib.TrgtHipError("trgt_read_meth: %d" % n)
    return None if n == -1 else [int(v) for v in out[:n]]


def _clip(fn, bases, quals, meth, cigar, ref_pos, a, b):
    bs, qs = np.frombuffer(bytes(bases), np.uint8), np.frombuffer(bytes(quals), np.uint8)
    me = None if meth is None else np.array(list(meth), np.uint8)
    w = cigar_words(cigar)
    ob, oq, om = np.zeros(len(bs) + 1, np.uint8), np.zeros(len(bs) + 1, np.uint8), np.zeros((len(me) if me is not None else 0) + 1, np.uint8)
    oc = np.zeros(len(w) + 2, np.uint32)
    nm, nc, rp = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    p = _lib.ptr
    n = fn(p(bs) if len(bs) else None, p(qs) if len(qs) else None, len(bs), p(me) if me is not None and len(me) else None, -1 if me is None else len(me),
           p(w) if len(w) else None, len(w), ref_pos, a, b, p(ob), p(oq), p(om), C.byref(nm), p(oc), C.byref(nc), C.byref(rp))
    if n < -1:
        raise _lib.TrgtHipError("clip: %d" % n)
    if n == -1:
        return None
    return dict(bases=bytes(ob[:n]), quals=bytes(oq[:n]), meth=None if nm.value < 0 else [int(v) for v in om[:nm.value]],
                cigar=cigar_text(oc[:nc.value]), ref_pos=int(rp.value))


def clip_to_region(bases, quals, meth, cigar, ref_pos, region):
    return _clip(_L().trgt_read_clip_to_region, bases, quals, meth, cigar, ref_pos, int(region[0]), int(region[1]))


def clip_bases(bases, quals, meth, cigar, ref_pos, left_len, right_len):
    return _clip(_L().trgt_read_clip_bases, bases, quals, meth, cigar, ref_pos, int(left_len), int(right_len))


def median(data):
    d = np.array(list(data), np.int32)
    out = C.c_float(0)
    rc = _L().trgt_median_i32(_lib.ptr(d) if len(d) else None, len(d), C.byref(out))
    if rc < 0:
        raise _lib.TrgtHipError("trgt_median_i32: %d" % rc)
    return None if rc == 0 else float(out.value)
