"""Host-side mirror of the reference's motif-HMM interface over the GPU batch ABI.

Reference (PacificBiosciences/trgt v3.0.0):
  build_hmm           src/hmm/builder.rs:4-78          -> build_hmm(motifs) -> Hmm
  Hmm::label          src/hmm/hmm_model.rs:144-156     -> Hmm.label(query)
  Annotation / Span   src/hmm/spans.rs:1-25            -> Annotation / Span
  label_with_hmm      src/trgt/workflows/tr.rs:454-492 -> label_with_hmm(motifs, seqs)
  count_motifs, collapse_labels, replace_invalid_bases  src/hmm/utils.rs:3-42
  encode_mc / encode_ms / encode_ap                     src/trgt/writers/write_vcf.rs:286-343
Every Viterbi fill / traceback / decode runs in trgt_hmm_batch (trgt_amd/csrc/hmm.hip); this module only
packs batches and unpacks results.
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib


@dataclass
class Span:  # spans.rs:1-6
    motif_index: int
    start: int
    end: int

    def __len__(self):
        return self.end - self.start


@dataclass
class Annotation:  # spans.rs:20-25
    labels: Optional[List[Span]]
    motif_counts: List[int]
    purity: float
    path: Optional[np.ndarray] = field(default=None, repr=False)


def _b(s):
    return s.encode() if isinstance(s, str) else bytes(s)


def replace_invalid_bases(seq, allowed):  # utils.rs:29-42
    seq, allowed = _b(seq), _b(allowed)
    return bytes(b if b in allowed else allowed[i % len(allowed)] for i, b in enumerate(seq))


def count_motifs(n_motifs, labels):  # utils.rs:3-9
    counts = [0] * n_motifs
    for s in labels:
        counts[s.motif_index] += 1
    return counts


def collapse_labels(spans):  # utils.rs:11-27
    out = []
    for s in spans:
        if out and out[-1].motif_index == s.motif_index and out[-1].end == s.start:
            out[-1] = Span(s.motif_index, out[-1].start, s.end)
        else:
            out.append(Span(s.motif_index, s.start, s.end))
    return out


def num_states(motifs):  # builder.rs:6
    return 7 + sum(3 * len(m) + 1 for m in motifs)


def pack_hmm_batch(motif_sets, jobs):
    """motif_sets: list of motif lists; jobs: list of (set_index, sequence).  Returns the ABI arrays (host)."""
    flat = [_b(m) for ms in motif_sets for m in ms]
    motif_off = np.zeros(len(flat) + 1, np.uint32)
    motif_off[1:] = np.cumsum([len(m) for m in flat])
    set_begin = np.zeros(len(motif_sets) + 1, np.uint32)
    set_begin[1:] = np.cumsum([len(ms) for ms in motif_sets])
    max_mlen = [max(len(m) for m in ms) for ms in motif_sets]
    seqs = [_b(s) for _, s in jobs]
    n = len(jobs)
    seq_len = np.array([len(s) for s in seqs], np.uint32)
    seq_off = np.zeros(n + 1, np.uint64)
    seq_off[1:] = np.cumsum(seq_len, dtype=np.uint64)
    job_set = np.array([j[0] for j in jobs], np.uint32)
    cap = np.array([_lib.lib().trgt_hmm_path_capacity(int(seq_len[i]), int(max_mlen[job_set[i]])) for i in range(n)], np.uint64)
    path_off = np.zeros(n + 1, np.uint64)
    path_off[1:] = np.cumsum(cap, dtype=np.uint64)
    span_off = np.zeros(n + 1, np.uint64)
    span_off[1:] = np.cumsum(seq_len.astype(np.uint64) + 1, dtype=np.uint64)
    nm = np.array([len(motif_sets[s]) for s in job_set], np.uint64)
    count_off = np.zeros(n + 1, np.uint64)
    count_off[1:] = np.cumsum(nm, dtype=np.uint64)
    blob = b"".join(seqs)
    return dict(motif_blob=np.frombuffer(b"".join(flat), np.uint8).copy(), motif_off=motif_off, set_motif_begin=set_begin,
                job_set=job_set, seq_blob=np.frombuffer(blob, np.uint8).copy() if blob else np.zeros(1, np.uint8),
                seq_off=seq_off, seq_len=seq_len, path_off=path_off, span_off=span_off, count_off=count_off,
                n_motifs=nm.astype(np.int64))


def models_check(motif_sets, ctx=None):
    """Bytes in which the device-built model tables differ from the host builder's (0 expected)."""
    ctx = ctx or _lib.context()
    b = pack_hmm_batch(motif_sets, [])
    n = C.c_int64(-1)
    p = _lib.ptr
    ctx.check(_lib.lib().trgt_hmm_models_check(ctx.handle, len(motif_sets), p(b["motif_blob"]), p(b["motif_off"]), p(b["set_motif_begin"]), C.byref(n)))
    return int(n.value)


def hmm_batch(batch, ctx=None, want_path=True, seq_blob_dev=None):
    """Run trgt_hmm_batch.  seq_blob_dev: optional torch uint8 tensor already resident in HBM."""
    ctx = ctx or _lib.context()
    n = len(batch["job_set"])
    path = np.zeros(int(batch["path_off"][-1]), np.uint16) if want_path else None
    path_len = np.zeros(n, np.uint32)
    spans = np.zeros(3 * int(batch["span_off"][-1]), np.int32)
    n_spans = np.zeros(n, np.uint32)
    counts = np.zeros(int(batch["count_off"][-1]), np.uint32)
    purity = np.zeros(n, np.float64)
    edit = np.zeros(n, np.int32)
    maxd = np.zeros(n, np.int32)
    p = _lib.ptr
    seq = seq_blob_dev if seq_blob_dev is not None else batch["seq_blob"]
    ctx.check(_lib.lib().trgt_hmm_batch(
        ctx.handle, len(batch["set_motif_begin"]) - 1, p(batch["motif_blob"]), p(batch["motif_off"]),
        p(batch["set_motif_begin"]), n, p(batch["job_set"]), p(seq), p(batch["seq_off"]), p(batch["seq_len"]), p(path),
        p(batch["path_off"]), p(path_len), p(spans), p(batch["span_off"]), p(n_spans), p(counts), p(batch["count_off"]),
        p(purity), p(edit), p(maxd)))
    return dict(path=path, path_len=path_len, spans=spans, n_spans=n_spans, counts=counts, purity=purity, edit=edit, maxd=maxd)


def unpack_annotations(batch, out, want_path=True):
    anns = []
    for j in range(len(batch["job_set"])):
        so, ns = int(batch["span_off"][j]), int(out["n_spans"][j])
        sp = out["spans"][3 * so:3 * (so + ns)].reshape(-1, 3)
        labels = [Span(int(a), int(b), int(c)) for a, b, c in sp] or None
        co, nm = int(batch["count_off"][j]), int(batch["n_motifs"][j])
        pth = None
        if want_path and out["path"] is not None:
            po = int(batch["path_off"][j])
            pth = out["path"][po:po + int(out["path_len"][j])].astype(np.int32)
        anns.append(Annotation(labels, [int(v) for v in out["counts"][co:co + nm]], float(out["purity"][j]), pth))
    return anns


class Hmm:
    """build_hmm(motifs) result; the model tables themselves are built inside the library per call."""

    def __init__(self, motifs, ctx=None):
        self.motifs_in = [replace_invalid_bases(m, b"ATCGN") for m in motifs]  # tr.rs:455-460
        self.num_states = num_states(self.motifs_in)
        self.ctx = ctx

    def label(self, query):  # hmm_model.rs:144-156 (query must be over ATCG, as in the reference)
        q = _b(query)
        if any(c not in b"ATCG" for c in q):
            raise ValueError("Hmm.label: query must be over ATCG (encode_base panics otherwise, hmm_model.rs:243-252)")
        return self.annotate([q])[0].path

    def annotate(self, seqs, want_path=True):
        batch = pack_hmm_batch([self.motifs_in], [(0, s) for s in seqs])
        return unpack_annotations(batch, hmm_batch(batch, self.ctx, want_path), want_path)


def build_hmm(motifs, ctx=None):  # builder.rs:4
    return Hmm(motifs, ctx)


def label_with_hmm(motifs, seqs, ctx=None):  # tr.rs:454-492
    return build_hmm(motifs, ctx).annotate(seqs, want_path=False)


# ---- VCF field encoders (write_vcf.rs:286-343)
def encode_mc(annotations):
    return ",".join("_".join(str(c) for c in a.motif_counts) for a in annotations)


def encode_ms(annotations):
    return ",".join("." if a.labels is None else "_".join("%d(%d-%d)" % (s.motif_index, s.start, s.end) for s in a.labels)
                    for a in annotations)


def encode_ap(annotations):
    return ",".join("." if math.isnan(a.purity) else "%.6f" % a.purity for a in annotations)
