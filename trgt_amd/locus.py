"""Host-side mirror of the reference's per-locus genotyper contract over the GPU batch ABI trgt_locus_batch.

Reference (PacificBiosciences/trgt v3.0.0):
  Params        src/trgt/workflows/tr.rs:17-22       -> Params
  Locus         src/trgt/locus.rs:13-23              -> the arrays of a batch (see pack / trgt_amd.synth.generate)
  analyze       src/trgt/workflows/tr.rs:24-109      -> analyze_batch (size / cluster genotyper, pre-clipped reads)
  LocusResult / Allele  workflows/locus_result.rs:6-23 -> LocusResult / Allele
  find_tr_spans src/trgt/genotype/span_locater.rs:32-68 -> find_tr_spans_batch
  encode_* of write_vcf.rs:286-377                   -> LocusResult.vcf_fields()
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .hmm import Annotation, Span, encode_ap, encode_mc, encode_ms


@dataclass
class Params:  # tr.rs:17-22 (+ --aln-scoring of cli.rs:271-280)
    min_flank_id_frac: float = 0.7
    search_flank_len: int = 250
    max_depth: int = 250
    aln_scoring: Tuple[int, int, int] = (2, 5, 1)
    host_threads: int = 0
    min_read_qual: float = 0.98   # --min-read-quality (cli.rs); < 0.9 switches filter_impure_trs on (tr.rs:37-50)


@dataclass
class Allele:  # locus_result.rs:6-12
    seq: bytes
    annotation: Annotation
    ci: Tuple[int, int]
    num_spanning: int
    meth: Optional[float] = None


@dataclass
class LocusResult:  # locus_result.rs:16-22
    genotype: List[Allele]
    reads: List[int]                 # indices of the kept spanning reads (into the locus' input reads), output order
    tr_spans: List[Tuple[int, int]]
    classification: List[int]

    def vcf_fields(self):  # write_vcf.rs:286-377
        g = self.genotype
        ann = [a.annotation for a in g]
        return dict(AL=",".join(str(len(a.seq)) for a in g), ALLR=",".join("%d-%d" % a.ci for a in g),
                    SD=",".join(str(a.num_spanning) for a in g), MC=encode_mc(ann), MS=encode_ms(ann), AP=encode_ap(ann))


def pack(loci):
    """loci: list of dict(left_flank, right_flank, tr, motifs, ploidy, reads[, genotyper ("size" | "cluster"), read_qual, and -- what
    genotype_flank reads -- hp_tag (None / 1 / 2 per read), start_offset, end_offset, mismatch_offsets (a list per read)]).
    Returns the ABI arrays (host)."""
    flank, tr, motifs, reads = bytearray(), bytearray(), bytearray(), bytearray()
    lf_off, lf_len, rf_off, rf_len, tr_off, tr_len, motif_off, set_begin, ploidy, lrb, read_off, read_len = ([] for _ in range(12))
    motif_off.append(0)
    set_begin.append(0)
    lrb.append(0)
    genotyper, read_qual = [], []
    has_meta = any(L.get("hp_tag") is not None or L.get("mismatch_offsets") is not None for L in loci)
    hp, so, eo, mm, mm_off = [], [], [], [], [0]
    for L in loci:
        if has_meta:
            n = len(L["reads"])
            hp += [-1 if v is None else int(v) for v in (L.get("hp_tag") if L.get("hp_tag") is not None else [None] * n)]
            so += list(L.get("start_offset") if L.get("start_offset") is not None else [0] * n)
            eo += list(L.get("end_offset") if L.get("end_offset") is not None else [0] * n)
            for lst in (L.get("mismatch_offsets") if L.get("mismatch_offsets") is not None else [[]] * n):
                mm += list(lst)
                mm_off.append(len(mm))
        genotyper.append(1 if L.get("genotyper", "size") in (1, "cluster") else 0)
        rq = L.get("read_qual")
        read_qual += [float("nan") if q is None else float(q) for q in (rq if rq is not None else [None] * len(L["reads"]))]
        lf_off.append(len(flank)); lf_len.append(len(L["left_flank"])); flank += L["left_flank"]
        rf_off.append(len(flank)); rf_len.append(len(L["right_flank"])); flank += L["right_flank"]
        tr_off.append(len(tr)); tr_len.append(len(L["tr"])); tr += L["tr"]
        for m in L["motifs"]:
            motifs += m if isinstance(m, bytes) else m.encode()
            motif_off.append(len(motifs))
        set_begin.append(len(motif_off) - 1)
        ploidy.append(L.get("ploidy", 2))
        for r in L["reads"]:
            read_off.append(len(reads)); read_len.append(len(r)); reads += r
        lrb.append(len(read_off))
    u8 = lambda b: np.frombuffer(bytes(b), np.uint8).copy() if len(b) else np.zeros(1, np.uint8)
    return dict(n_loci=len(loci), n_reads=len(read_off), flank_blob=u8(flank), lf_off=np.array(lf_off, np.uint64),
                lf_len=np.array(lf_len, np.uint32), rf_off=np.array(rf_off, np.uint64), rf_len=np.array(rf_len, np.uint32),
                tr_blob=u8(tr), tr_off=np.array(tr_off, np.uint64), tr_len=np.array(tr_len, np.uint32), motif_blob=u8(motifs),
                motif_off=np.array(motif_off, np.uint32), set_motif_begin=np.array(set_begin, np.uint32),
                ploidy=np.array(ploidy, np.uint8), locus_read_begin=np.array(lrb, np.uint64), read_blob=u8(reads),
                read_off=np.array(read_off, np.uint64), read_len=np.array(read_len, np.uint32),
                genotyper=np.array(genotyper, np.uint8), read_qual=np.array(read_qual, np.float64),
                **(dict(hp_tag=np.array(hp, np.int16), start_offset=np.array(so, np.int32), end_offset=np.array(eo, np.int32),
                        mismatch_offsets=np.array(mm + [0], np.int32), mismatch_off=np.array(mm_off, np.uint64)) if has_meta else {}))


def find_tr_spans_batch(batch, params=Params(), ctx=None, flank_dev=None, reads_dev=None):
    """span_locater.rs:32-68 for every read of the batch.  *_dev: optional torch uint8 tensors already in HBM."""
    ctx = ctx or _lib.context()
    sp = _lib.SpanParams(params.search_flank_len, params.min_flank_id_frac, *params.aln_scoring)
    n = int(batch["n_reads"])
    ss, se = np.zeros(n, np.int32), np.zeros(n, np.int32)
    lh, rh = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    p = _lib.ptr
    ctx.check(_lib.lib().trgt_find_spans_batch(
        ctx.handle, C.byref(sp), int(batch["n_loci"]), p(flank_dev if flank_dev is not None else batch["flank_blob"]),
        p(batch["lf_off"]), p(batch["lf_len"]), p(batch["rf_off"]), p(batch["rf_len"]), p(batch["locus_read_begin"]),
        p(reads_dev if reads_dev is not None else batch["read_blob"]), p(batch["read_off"]), p(batch["read_len"]), p(ss), p(se),
        p(lh), p(rh)))
    return ss, se, lh, rh


class BatchOutputs:
    """Caller-owned output buffers of trgt_locus_batch, reusable across calls of the same batch shape."""

    def __init__(self, batch):
        nl, nr = int(batch["n_loci"]), int(batch["n_reads"])
        lrb = batch["locus_read_begin"]
        rl = batch["read_len"]
        # longest read of every locus + 8 (vectorised: a Python loop over the loci was 10 ms per 1 000-locus chunk -- most of what the
        # end-to-end leg of bench.py reported as its GPU stage)
        cap = np.full(nl, 8, np.uint32)
        if nl > 0 and nr > 0:
            starts = np.asarray(lrb[:-1], np.int64)
            nonempty = np.asarray(lrb[1:], np.int64) > starts
            if nonempty.any():
                # (segments of consecutive non-empty loci: the empty ones in between own no reads; read_len is cut at the batch's read count,
                #  reduceat takes its last segment to the END of the array, and a padded / reused array is longer than that)
                seg_max = np.maximum.reduceat(np.asarray(rl[:int(lrb[nl])], np.uint32), starts[nonempty])
                cap[nonempty] = seg_max + 8
        self.allele_cap = cap
        cap2 = np.repeat(cap.astype(np.uint64), 2)
        self.allele_off = np.zeros(2 * nl, np.uint64)
        self.allele_off[1:] = np.cumsum(cap2[:-1])
        self.allele_blob = np.zeros(int(cap2.sum()) + 8, np.uint8)
        self.allele_len = np.zeros(2 * nl, np.uint32)
        self.span_off = np.zeros(2 * nl, np.uint64)
        self.span_off[1:] = np.cumsum(cap2[:-1] + 1)
        self.spans3 = np.zeros(3 * (int((cap2 + 1).sum()) + 8), np.int32)
        self.n_spans = np.zeros(2 * nl, np.uint32)
        nm = np.diff(batch["set_motif_begin"]).astype(np.uint64)
        nm2 = np.repeat(nm, 2)
        self.count_off = np.zeros(2 * nl, np.uint64)
        self.count_off[1:] = np.cumsum(nm2[:-1])
        self.motif_counts = np.zeros(int(nm2.sum()) + 8, np.uint32)
        self.n_motifs = nm.astype(np.int64)
        self.purity = np.zeros(2 * nl, np.float64)
        self.span_start, self.span_end = np.zeros(nr, np.int32), np.zeros(nr, np.int32)
        self.n_alleles = np.zeros(nl, np.int32)
        self.ci = np.zeros(4 * nl, np.int32)
        self.num_spanning = np.zeros(2 * nl, np.int32)
        self.classification, self.read_rank = np.zeros(nr, np.int32), np.zeros(nr, np.int32)
        self.stats = np.zeros(24, np.int64)
        self.gt_size, self.flipped = np.zeros(2 * nl, np.int32), np.zeros(nl, np.uint8)
        p = _lib.ptr
        self.c_out = _lib.LocusBatchOut(*[p(getattr(self, n)).value for n in (
            "span_start", "span_end", "n_alleles", "allele_blob", "allele_off", "allele_cap", "allele_len", "ci", "num_spanning",
            "classification", "read_rank", "spans3", "span_off", "n_spans", "motif_counts", "count_off", "purity", "stats", "gt_size", "flipped")])


_CIN_KEYS = ("lf_off", "lf_len", "rf_off", "rf_len", "tr_blob", "tr_off", "tr_len", "motif_blob", "motif_off", "set_motif_begin", "ploidy",
             "locus_read_begin", "read_off", "read_len", "genotyper", "read_qual", "hp_tag", "start_offset", "end_offset", "mismatch_offsets", "mismatch_off")


def _batch_in(batch, flank, reads):
    p = _lib.ptr
    # The input struct of a batch is rebuilt only when one of its arrays is replaced (20 pointer conversions per call otherwise).
    # The cache entry HOLDS the arrays it was built from and compares them by identity: an id() alone could be reused by a new
    # array allocated at a freed one's address, and the struct would then carry a dangling pointer.
    key = (int(batch["n_loci"]), flank, reads) + tuple(batch.get(k) for k in _CIN_KEYS) + (int(batch.get("read_encoding", 0)),)
    cached = batch.get("_cin")
    if cached is None or len(cached[0]) != len(key) or cached[0][0] != key[0] or any(a is not b for a, b in zip(cached[0][1:], key[1:])):
        cin = _lib.LocusBatchIn(int(batch["n_loci"]), *[p(v).value for v in (
            flank, batch["lf_off"], batch["lf_len"], batch["rf_off"],
            batch["rf_len"], batch["tr_blob"], batch["tr_off"], batch["tr_len"], batch["motif_blob"], batch["motif_off"],
            batch["set_motif_begin"], batch["ploidy"], batch["locus_read_begin"],
            reads, batch["read_off"], batch["read_len"])],
            *[p(batch.get(k)).value if batch.get(k) is not None else None for k in ("genotyper", "read_qual", "hp_tag", "start_offset", "end_offset",
                                                                                     "mismatch_offsets", "mismatch_off")],
            int(batch.get("read_encoding", 0)))
        cached = (key, cin)
        batch["_cin"] = cached
    return cached[1]


def pack_bam4(batch, pinned=False):
    """The batch with its reads as BAM 4-bit codes (TRGT_READS_BAM4, include/trgt_hip.h): a shallow copy whose read_blob / read_off are the
    packed ones (trgt_reads_pack_bam4) and read_encoding = 1.  pinned: the packed blob in pinned host memory (a torch tensor kept in the
    dict).  read_len and everything else are shared with the original."""
    nr = int(batch["locus_read_begin"][int(batch["n_loci"])])
    total = int(((batch["read_len"][:nr].astype(np.uint64) + 1) // 2).sum())
    if pinned:
        import torch
        t = torch.empty(max(total, 1), dtype=torch.uint8).pin_memory()
        packed = t.numpy()
    else:
        t, packed = None, np.zeros(max(total, 1), np.uint8)
    off = np.zeros(max(nr, 1), np.uint64)
    blob = np.ascontiguousarray(batch["read_blob"])
    got = _lib.lib().trgt_reads_pack_bam4(_lib.ptr(blob), nr, _lib.ptr(batch["read_off"]), _lib.ptr(batch["read_len"]), _lib.ptr(packed), _lib.ptr(off))
    if got != total:
        raise RuntimeError(f"trgt_reads_pack_bam4 returned {got}, expected {total}")
    b = {k: v for k, v in batch.items() if k != "_cin"}
    b["read_blob"], b["read_off"], b["read_encoding"] = packed, off, 1
    if t is not None:
        b["_read_blob_pinned"] = t
    return b


def _locus_params(params):
    return _lib.LocusParams(params.search_flank_len, params.min_flank_id_frac, params.max_depth, params.aln_scoring[0],
                            params.aln_scoring[1], params.aln_scoring[2], params.host_threads, params.min_read_qual)


def run_batch(batch, params=Params(), ctx=None, outputs=None, flank_dev=None, reads_dev=None):
    """trgt_locus_batch on a packed batch.  Returns the (reusable) BatchOutputs."""
    ctx = ctx or _lib.context()
    out = outputs or BatchOutputs(batch)
    flank, reads = flank_dev if flank_dev is not None else batch["flank_blob"], reads_dev if reads_dev is not None else batch["read_blob"]
    cin = _batch_in(batch, flank, reads)
    lp = _locus_params(params)
    ctx.check(_lib.lib().trgt_locus_batch(ctx.handle, C.byref(lp), C.byref(cin), C.byref(out.c_out)))
    return out


def run_many(pool, batches, params=Params(), outputs=None, flank_dev=None, reads_dev=None, out_per_context=False):
    """trgt_locus_batch_many: the batches (packed dicts) through the contexts of a _lib.Pool, one worker thread per context inside the
    library.  outputs: one BatchOutputs per batch, or -- out_per_context -- one per context (every batch a context runs overwrites its
    entry: throughput measurements).  flank_dev / reads_dev: HBM copies shared by all batches (same device).  Returns (outputs, ran_on)."""
    if outputs is None:
        outputs = [BatchOutputs(batches[i if not out_per_context else 0]) for i in range(pool.n if out_per_context else len(batches))]
    cins = []
    for b in batches:
        flank, reads = flank_dev if flank_dev is not None else b["flank_blob"], reads_dev if reads_dev is not None else b["read_blob"]
        cins.append(_batch_in(b, flank, reads))
    lp = _locus_params(params)
    ran = pool.run_many(lp, cins, [o.c_out for o in outputs], out_per_context)
    return outputs, ran


class Ticket:
    """A batch handed to trgt_locus_batch_submit: keeps everything the library still points at alive until wait()."""

    def __init__(self, ctx, batch, out, cin, lp, blobs, ticket):
        self.ctx, self.batch, self.outputs, self._cin, self._lp, self._blobs, self.id = ctx, batch, out, cin, lp, blobs, ticket

    def wait(self):
        """trgt_locus_batch_wait: analyses the batch (its reads were uploading meanwhile) and returns the BatchOutputs."""
        self.ctx.check(_lib.lib().trgt_locus_batch_wait(self.ctx.handle, self.id))
        return self.outputs


def submit_batch(batch, params=Params(), ctx=None, outputs=None, flank=None, reads=None):
    """trgt_locus_batch_submit: start uploading the batch's read / flank bytes (pass pinned torch tensors or numpy arrays in
    `reads` / `flank` for an asynchronous copy) and return a Ticket; ticket.wait() runs the analysis.  At most two tickets may be
    outstanding per context, waited for in order: submit(b0); loop { submit(b[k+1]); wait(b[k]) } overlaps upload and compute."""
    ctx = ctx or _lib.context()
    out = outputs or BatchOutputs(batch)
    fl, rd = flank if flank is not None else batch["flank_blob"], reads if reads is not None else batch["read_blob"]
    # (a struct of its own: the cached one of run_batch may be rebuilt while this batch is in flight)
    p = _lib.ptr
    cin = _lib.LocusBatchIn(int(batch["n_loci"]), *[p(v).value for v in (
        fl, batch["lf_off"], batch["lf_len"], batch["rf_off"], batch["rf_len"], batch["tr_blob"], batch["tr_off"], batch["tr_len"],
        batch["motif_blob"], batch["motif_off"], batch["set_motif_begin"], batch["ploidy"], batch["locus_read_begin"],
        rd, batch["read_off"], batch["read_len"])],
        *[p(batch.get(k)).value if batch.get(k) is not None else None for k in ("genotyper", "read_qual", "hp_tag", "start_offset", "end_offset",
                                                                                 "mismatch_offsets", "mismatch_off")],
        int(batch.get("read_encoding", 0)))
    # (the same optional-field list as _batch_in: without hp_tag / offsets the library would skip genotype_flank, tr.rs:69-75)
    lp = _locus_params(params)
    t = C.c_int64(0)
    ctx.check(_lib.lib().trgt_locus_batch_submit(ctx.handle, C.byref(lp), C.byref(cin), C.byref(out.c_out), C.byref(t)))
    return Ticket(ctx, batch, out, cin, lp, (fl, rd) + tuple(batch.get(k) for k in _CIN_KEYS), t.value)


def locus_result(batch, out, l):
    """Unpack locus l of a finished batch into the reference's LocusResult shape."""
    a0, a1 = int(batch["locus_read_begin"][l]), int(batch["locus_read_begin"][l + 1])
    geno = []
    for a in range(int(out.n_alleles[l])):
        s = 2 * l + a
        seq = bytes(out.allele_blob[int(out.allele_off[s]):int(out.allele_off[s]) + int(out.allele_len[s])])
        so, ns = int(out.span_off[s]), int(out.n_spans[s])
        labels = [Span(int(x), int(y), int(z)) for x, y, z in out.spans3[3 * so:3 * (so + ns)].reshape(-1, 3)] or None
        co = int(out.count_off[s])
        counts = [int(v) for v in out.motif_counts[co:co + int(out.n_motifs[l])]]
        geno.append(Allele(seq, Annotation(labels, counts, float(out.purity[s])), (int(out.ci[4 * l + 2 * a]), int(out.ci[4 * l + 2 * a + 1])),
                           int(out.num_spanning[2 * l + a])))
    rank = out.read_rank[a0:a1]
    kept = [int(i) for i in np.argsort(np.where(rank >= 0, rank, 1 << 30), kind="stable")[:int((rank >= 0).sum())]]
    return LocusResult(geno, kept, [(int(out.span_start[a0 + i]), int(out.span_end[a0 + i])) for i in kept],
                       [int(out.classification[a0 + i]) for i in kept])


def analyze_batch(loci, params=Params(), ctx=None):
    """analyze() for a list of loci (dicts, see pack) -> list of LocusResult."""
    batch = pack(loci)
    out = run_batch(batch, params, ctx)
    return [locus_result(batch, out, l) for l in range(len(loci))]
