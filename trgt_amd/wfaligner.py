"""Host-side mirror of the reference's WFAligner wrapper (PacificBiosciences/trgt v3.0.0 src/wfaligner.rs) over
the GPU batch ABI trgt_wfa_batch.

Same names and argument meaning as the Rust wrapper:
  MemoryModel / AlignmentScope / Heuristic / AlignmentStatus / Penalties   wfaligner.rs:4-159
  WFAlignerBuilder (.indel/.edit/.linear/.affine/.affine2p/.with_heuristic/.build)   :247-380
  WFAligner.align_end_to_end / align_ends_free / score / cigar_operations / get_sam_cigar /
  decode_sam_cigar / count_matches / get_alignment_span / get_alignment / cigar_score /
  cigar_score_clipped / set_heuristic / get_heuristics / get_penalties               :386-1126
plus the batch forms (align_end_to_end_batch / align_ends_free_batch) that the GPU actually wants.
Accessors that panic in the reference raise here (same conditions).  All alignments run on the GPU.
"""
import ctypes as C
from dataclasses import dataclass
from enum import Enum, IntEnum
from typing import List, Tuple

import numpy as np

from . import _lib

I32_MIN = -2147483648


class MemoryModel(IntEnum):  # wfaligner.rs:4-10
    MemoryHigh = 0
    MemoryMed = 1
    MemoryLow = 2
    MemoryUltraLow = 3


class AlignmentScope(IntEnum):  # :52-56
    Score = 0
    Alignment = 1


class AlignmentStatus(IntEnum):  # :125-134 (WF_STATUS_*)
    StatusAlgCompleted = 0
    StatusAlgPartial = 1
    StatusMaxStepsReached = -100
    StatusOOM = -200
    StatusUnattainable = -300


class DistanceMetric(IntEnum):  # :79-85
    Indel = 0
    Edit = 1
    GapLinear = 2
    GapAffine = 3
    GapAffine2p = 4


class WfaOp(Enum):  # :12-18
    Match = "M"
    Subst = "X"
    Ins = "I"
    Del = "D"


@dataclass(frozen=True)
class Heuristic:  # :68-77
    kind: str
    args: Tuple[int, ...] = ()

    @staticmethod
    def none():
        return Heuristic("None")

    @staticmethod
    def wfadaptive(min_wavefront_length, max_distance_threshold, steps_between_cutoffs):
        return Heuristic("WFadaptive", (min_wavefront_length, max_distance_threshold, steps_between_cutoffs))

    @staticmethod
    def wfmash(a, b, c):
        return Heuristic("WFmash", (a, b, c))

    @staticmethod
    def xdrop(a, b):
        return Heuristic("XDrop", (a, b))

    @staticmethod
    def zdrop(a, b):
        return Heuristic("ZDrop", (a, b))

    @staticmethod
    def banded_static(a, b):
        return Heuristic("BandedStatic", (a, b))

    @staticmethod
    def banded_adaptive(a, b, c):
        return Heuristic("BandedAdaptive", (a, b, c))


@dataclass(frozen=True)
class Penalties:  # :87-110
    kind: str
    match_: int = 0
    mismatch: int = 0
    indel: int = 0
    gap_opening: int = 0
    gap_extension: int = 0
    gap_opening2: int = 0
    gap_extension2: int = 0


@dataclass
class WfaAlign:  # :32-50
    score: int
    ystart: int
    xstart: int
    yend: int
    xend: int
    ylen: int
    xlen: int
    operations: List[WfaOp]


class WFAlignerBuilder:  # :247-380
    def __init__(self, alignment_scope, memory_model):
        self.scope, self.memory = AlignmentScope(alignment_scope), MemoryModel(memory_model)
        self.pen = None
        self.heuristic = None  # None = keep wavefront_aligner_attr_default's wfadaptive(10,50,1)

    def indel(self):
        self.pen = Penalties("Indel")
        return self

    def edit(self):
        self.pen = Penalties("Edit")
        return self

    def linear(self, mismatch, indel):
        return self.linear_with_match(0, mismatch, indel)

    def linear_with_match(self, match_, mismatch, indel):
        self.pen = Penalties("Linear", match_=match_, mismatch=mismatch, indel=indel)
        return self

    def affine(self, mismatch, gap_opening, gap_extension):
        return self.affine_with_match(0, mismatch, gap_opening, gap_extension)

    def affine_with_match(self, match_, mismatch, gap_opening, gap_extension):
        self.pen = Penalties("Affine", match_=match_, mismatch=mismatch, gap_opening=gap_opening, gap_extension=gap_extension)
        return self

    def affine2p(self, mismatch, o1, e1, o2, e2):
        return self.affine2p_with_match(0, mismatch, o1, e1, o2, e2)

    def affine2p_with_match(self, match_, mismatch, o1, e1, o2, e2):
        self.pen = Penalties("Affine2p", match_=match_, mismatch=mismatch, gap_opening=o1, gap_extension=e1, gap_opening2=o2,
                             gap_extension2=e2)
        return self

    def with_heuristic(self, heuristic):
        self.heuristic = heuristic
        return self

    def build(self, ctx=None):
        if self.pen is None:
            raise RuntimeError("Must set a penalty model before building the aligner")  # :361-363
        return WFAligner(self.scope, self.memory, self.pen, self.heuristic, ctx)


class WFAligner:
    def __init__(self, scope, memory, pen, heuristic, ctx=None):
        self.scope, self.memory, self.pen = scope, memory, pen
        self.heuristic = heuristic if heuristic is not None else Heuristic.wfadaptive(10, 50, 1)
        self.ctx = ctx
        self._last = None  # result of the most recent single alignment (the aligner "owns" its CIGAR, :386-389)
        self._span = "end2end"

    @staticmethod
    def builder(alignment_scope, memory_model):
        return WFAlignerBuilder(alignment_scope, memory_model)

    # ---- configuration accessors
    def get_penalties(self):
        return self.pen

    def get_heuristics(self):
        return self.heuristic

    def set_heuristic(self, heuristic):
        self.heuristic = heuristic

    @property
    def metric(self):
        return DistanceMetric[{"Indel": "Indel", "Edit": "Edit", "Linear": "GapLinear", "Affine": "GapAffine",
                               "Affine2p": "GapAffine2p"}[self.pen.kind]]

    def _params(self, span, pbf=0, pef=0, tbf=0, tef=0):
        if self.pen.match_ != 0:  # the builder and the accessors take them (wfaligner.rs:1510-1521); no kernel aligns with them
            raise NotImplementedError("non-zero match scores are never used on TRGT's paths (wfaligner.rs:292-294)")
        p = _lib.WfaParams()
        _lib.lib().trgt_wfa_default_params(C.byref(p))
        p.metric = int(self.metric)
        if self.pen.kind == "Linear":
            p.mismatch, p.gap_ext1 = self.pen.mismatch, self.pen.indel
        elif self.pen.kind in ("Affine", "Affine2p"):
            p.mismatch, p.gap_open1, p.gap_ext1 = self.pen.mismatch, self.pen.gap_opening, self.pen.gap_extension
            p.gap_open2, p.gap_ext2 = self.pen.gap_opening2, self.pen.gap_extension2
        p.span = 1 if span == "endsfree" else 0
        p.pattern_begin_free, p.pattern_end_free, p.text_begin_free, p.text_end_free = pbf, pef, tbf, tef
        p.scope = int(self.scope)
        p.memory_mode = int(self.memory)
        if self.heuristic.kind == "None":
            p.heuristic = 0
        elif self.heuristic.kind == "WFadaptive":
            p.heuristic = 1
            p.h_min_wavefront_length, p.h_max_distance_threshold, p.h_steps_between_cutoffs = self.heuristic.args
        else:
            # the rest of the enum by the header's numbers (include/trgt_hip.h): the library answers TRGT_ERR_UNSUPPORTED for the whole
            # batch and touches no output -- the documented contract for the heuristics no call site of the genotype path uses
            p.heuristic = {"WFmash": 2, "XDrop": 3, "ZDrop": 4, "BandedStatic": 5, "BandedAdaptive": 6}.get(self.heuristic.kind, 99)
        return p

    # ---- batch forms
    def _run_batch(self, p, patterns, texts, want_ops=True):
        ctx = self.ctx or _lib.context()
        n = len(patterns)
        pats = [bytes(x) for x in patterns]
        txts = [bytes(x) for x in texts]
        plen = np.array([len(x) for x in pats], np.uint32)
        tlen = np.array([len(x) for x in txts], np.uint32)
        blob = b"".join(pats) + b"".join(txts)
        pat_off = np.zeros(n, np.uint64)
        pat_off[1:] = np.cumsum(plen[:-1], dtype=np.uint64)
        txt_off = np.zeros(n, np.uint64)
        txt_off[1:] = np.cumsum(tlen[:-1], dtype=np.uint64)
        txt_off += np.uint64(int(plen.sum()))
        seqs = np.frombuffer(blob, np.uint8).copy() if blob else np.zeros(1, np.uint8)
        cap = plen.astype(np.uint64) + tlen.astype(np.uint64) + 1
        coff = np.zeros(n + 1, np.uint64)
        coff[1:] = np.cumsum(cap, dtype=np.uint64)
        status = np.zeros(n, np.int32)
        score = np.zeros(n, np.int32)
        n_match = np.zeros(n, np.int32)
        span4 = np.zeros(4 * n, np.uint32)
        cigar = np.zeros(int(coff[-1]), np.uint32)
        clen = np.zeros(n, np.uint32)
        ops = np.zeros(int(coff[-1]), np.uint8) if want_ops else None
        olen = np.zeros(n, np.uint32)
        q = _lib.ptr
        ctx.check(_lib.lib().trgt_wfa_batch(ctx.handle, C.byref(p), n, q(seqs), q(pat_off), q(plen), q(txt_off), q(tlen),
                                            q(status), q(score), q(n_match), q(span4), q(cigar), q(coff), q(clen), q(ops),
                                            q(coff) if want_ops else None, q(olen) if want_ops else None))
        return dict(status=status, score=score, n_match=n_match, span4=span4.reshape(-1, 4), cigar=cigar, cigar_off=coff,
                    cigar_len=clen, ops=ops, ops_len=olen, plen=plen, tlen=tlen)

    def align_end_to_end_batch(self, patterns, texts, want_ops=True):
        return self._run_batch(self._params("end2end"), patterns, texts, want_ops)

    def align_ends_free_batch(self, patterns, pattern_begin_free, pattern_end_free, texts, text_begin_free, text_end_free,
                              want_ops=True):
        """Free-end lengths are scalars applied to every job; -1 means "the sequence length" (TRGT's flank call)."""
        return self._run_batch(self._params("endsfree", pattern_begin_free, pattern_end_free, text_begin_free, text_end_free),
                               patterns, texts, want_ops)

    # ---- the reference's single-alignment surface
    def _keep(self, r, span, plen, tlen):
        n = int(r["ops_len"][0])
        self._last = dict(status=int(r["status"][0]), score=int(r["score"][0]), ops=bytes(r["ops"][:n]),
                          n_match=int(r["n_match"][0]), span=[int(v) for v in r["span4"][0]],
                          cigar=[int(v) for v in r["cigar"][:int(r["cigar_len"][0])]], plen=plen, tlen=tlen)
        self._span = span
        return AlignmentStatus(self._last["status"])

    def align_end_to_end(self, pattern, text):  # :489-501
        return self._keep(self.align_end_to_end_batch([pattern], [text]), "end2end", len(pattern), len(text))

    def align_ends_free(self, pattern, pattern_begin_free, pattern_end_free, text, text_begin_free, text_end_free):  # :503-528
        r = self.align_ends_free_batch([pattern], pattern_begin_free, pattern_end_free, [text], text_begin_free, text_end_free)
        return self._keep(r, "endsfree", len(pattern), len(text))

    def score(self):  # :530-532
        return self._last["score"]

    def _need_alignment(self, what):
        if self.scope == AlignmentScope.Score:
            raise RuntimeError("Cannot %s when AlignmentScope is Score" % what)

    def cigar_operations(self):  # :910-929
        return b"" if self.scope == AlignmentScope.Score else self._last["ops"]

    def get_sam_cigar(self, show_mismatches):  # :932-959
        self._need_alignment("get SAM CIGAR")
        if show_mismatches:
            return list(self._last["cigar"])
        out = []
        for e in self._last["cigar"]:
            code = e & 0xF
            code = 0 if code in (7, 8) else code
            if out and (out[-1] & 0xF) == code:
                out[-1] += (e >> 4) << 4
            else:
                out.append(((e >> 4) << 4) | code)
        return out

    @staticmethod
    def decode_sam_cigar(buf):  # :961-984
        return [(int(e) >> 4, "MIDNSHP=X"[int(e) & 0xF] if (int(e) & 0xF) <= 8 else "?") for e in buf]

    def count_matches(self):  # :988-1000
        self._need_alignment("count matches")
        return self._last["n_match"]

    def get_alignment_span(self):  # :864-908
        s = self._last["span"]
        return (s[0], s[1]), (s[2], s[3])

    def get_alignment(self):  # :784-860
        s = self._last["span"]
        return WfaAlign(self._last["score"], s[2], s[0], s[3], s[1], self._last["tlen"], self._last["plen"],
                        [WfaOp(chr(c)) for c in self._last["ops"]])

    def _op_score(self, op, n):  # :534-593
        k = self.pen.kind
        if k in ("Indel", "Edit"):
            return 0 if op == "M" else n
        if op == "M":
            return n * self.pen.match_
        if op == "X":
            return n * self.pen.mismatch
        if k == "Linear":
            return n * self.pen.indel
        s1 = self.pen.gap_opening + self.pen.gap_extension * n
        return s1 if k == "Affine" else min(s1, self.pen.gap_opening2 + self.pen.gap_extension2 * n)

    def _score_range(self, b, e):
        ops = self._last["ops"].decode()
        if b >= e:
            return 0
        score, i = 0, b
        while i < e:
            j = i
            while j < e and ops[j] == ops[i]:
                j += 1
            s = self._op_score(ops[i], j - i)
            score += s if self.pen.kind in ("Indel", "Edit") else -s
            i = j
        return score

    def cigar_score(self):  # :1002-1026
        self._need_alignment("calculate CIGAR score")
        return self._score_range(0, len(self._last["ops"]))

    def cigar_score_clipped(self, flank_len):  # :595-705
        self._need_alignment("clip")
        n = len(self._last["ops"])
        b = flank_len
        return self._score_range(b, max(b, n - flank_len))

    def cigar_string(self, flank_len=None):  # :1029-1061 (test helper in the reference)
        ops = self._last["ops"].decode()
        f = flank_len or 0
        ops = ops[f:len(ops) - f] if f else ops
        out, i = [], 0
        while i < len(ops):
            j = i
            while j < len(ops) and ops[j] == ops[i]:
                j += 1
            out.append("%d%s" % (j - i, ops[i]))
            i = j
        return "".join(out)


def flank_filter_batch(patterns, texts, min_matches, scoring=(2, 5, 1), ctx=None, early_reject=False):
    """trgt_flank_filter_batch: the pre-filter trgt_find_spans_batch runs in front of the back-tracing kernel
    (span_locater.rs:14-22).  Per job: the exact optimal score of align_ends_free(pattern, 0, 0, text, |text|, |text|),
    an upper bound on count_matches() of the reference's alignment, keep = bound >= min_matches (or "not judged").
    early_reject: give an alignment up once no cell of its wavefronts can reach min_matches (score INT32_MIN + 1 then).
    Returns dict(score, bound, keep, offsets)."""
    ctx = ctx or _lib.context()
    n = len(patterns)
    pats = [bytes(x) for x in patterns]
    txts = [bytes(x) for x in texts]
    plen = np.array([len(x) for x in pats], np.uint32)
    tlen = np.array([len(x) for x in txts], np.uint32)
    blob = b"".join(pats) + b"".join(txts)
    pat_off = np.zeros(n, np.uint64)
    pat_off[1:] = np.cumsum(plen[:-1], dtype=np.uint64)
    txt_off = np.zeros(n, np.uint64)
    txt_off[1:] = np.cumsum(tlen[:-1], dtype=np.uint64)
    txt_off += np.uint64(int(plen.sum()))
    seqs = np.frombuffer(blob, np.uint8).copy() if blob else np.zeros(1, np.uint8)
    sp = _lib.SpanParams()
    sp.flank_len, sp.min_flank_id_frac = 0, 0.0
    sp.mism, sp.gapo, sp.gape = scoring
    score = np.zeros(n, np.int32)
    bound = np.zeros(n, np.int32)
    keep = np.zeros(n, np.uint8)
    offsets = C.c_int64(0)
    q = _lib.ptr
    ctx.check(_lib.lib().trgt_flank_filter_batch(ctx.handle, C.byref(sp), n, q(seqs), q(pat_off), q(plen), q(txt_off), q(tlen),
                                                 int(min_matches), 1 if early_reject else 0, q(score), q(bound), q(keep), C.byref(offsets)))
    return dict(score=score, bound=bound, keep=keep, offsets=offsets.value)
