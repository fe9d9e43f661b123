"""Large seeded fuzz of trgt_wfa_batch against the CPU oracle: every distance metric, end-to-end / ends-free shapes, score / alignment
scope, the memory modes (BiWFA included), heuristics.  Compares status, score, count_matches, spans, CIGAR runs and op strings of
every job.  A developer tool (the GPU suite keeps small batches of the same shapes: tests/test_wfa_gpu.py).

    python tests/tools/wfa_fuzz.py [jobs_per_mode=20000] [seed=1]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from trgt_amd import wfaligner as W
from oracle import binding as oracle

LUT = np.frombuffer(b"ACGT", np.uint8)


def rand_idx(rng, n):
    return rng.integers(0, 4, n, dtype=np.uint8)


def mutate_idx(rng, a, sub, ins, dele):
    if len(a) == 0:
        return a
    a = a[rng.random(len(a)) >= dele]
    s = rng.random(len(a)) < sub
    a = np.where(s, (a + rng.integers(1, 4, len(a), dtype=np.uint8)) % 4, a).astype(np.uint8)
    k = (rng.random(len(a)) < ins).astype(np.int64)
    out = np.repeat(a, 1 + k)
    pos = np.cumsum(1 + k) - 1          # last copy of each base that got an insertion after it
    ins_at = pos[k > 0]
    out[ins_at] = rand_idx(rng, len(ins_at))
    return out


def tobytes(a):
    return LUT[a].tobytes()


def gen_pairs(rng, n, kind):
    pats, txts = [], []
    for i in range(n):
        if kind == "flank":        # 250-bp piece vs read; some truncated / absent
            f = rand_idx(rng, 250)
            noisy = i % 4 == 0
            m = mutate_idx(rng, f, *((0.05, 0.02, 0.02) if noisy else (0.004, 0.002, 0.002)))
            left, right = rand_idx(rng, int(rng.integers(0, 500))), rand_idx(rng, int(rng.integers(0, 700)))
            r = np.concatenate([left, m, right])
            if i % 10 == 0:
                cut = int(rng.integers(1, 250))
                r = np.concatenate([m[cut:], right]) if i % 20 == 0 else np.concatenate([left, m[:cut]])
            if i % 33 == 0:
                r = rand_idx(rng, int(rng.integers(1, 300)))
            a, b = f, r
        elif kind == "str":        # STR allele vs noisy copy with stutter
            mot = rand_idx(rng, int(rng.integers(2, 7)))
            L = int(rng.integers(10, 500))
            a = np.tile(mot, L // len(mot) + 1)[:L]
            b = mutate_idx(rng, a, 0.01, 0.01, 0.01)
            if i % 3 == 0:
                k = len(mot) * int(rng.integers(1, 4))
                b = np.concatenate([b[:len(b) // 2], np.tile(mot, 3)[:k], b[len(b) // 2:]]) if i % 2 else b[k:]
        elif kind == "consensus":  # a read of an allele against the central read of its cluster: HiFi errors, a stutter now and then, long alleles too
            mot = rand_idx(rng, int(rng.integers(2, 9)))
            L = int(rng.integers(20, 2500 if i % 16 == 0 else 700))
            a = np.tile(mot, L // len(mot) + 1)[:L]
            if i % 3 == 0:   # an interruption or two inside the repeat
                for _ in range(int(rng.integers(1, 3))):
                    at = int(rng.integers(0, len(a)))
                    a = np.concatenate([a[:at], rand_idx(rng, int(rng.integers(1, 12))), a[at:]])
            e = (0.001, 0.0005, 0.0005) if i % 4 else (0.01, 0.006, 0.006)
            b = mutate_idx(rng, a, *e)
            if i % 5 == 0:
                k = len(mot) * int(rng.integers(1, 3))
                at = int(rng.integers(0, len(b) + 1))
                b = np.concatenate([b[:at], np.tile(mot, 3)[:k], b[at:]]) if i % 2 else np.concatenate([b[:at], b[at + k:]])
            if i % 7 == 0:
                a = mutate_idx(rng, a, *e)
            if i % 13 == 0:
                a, b = b, a
        elif kind == "alleles":    # reads of two alleles of one locus against each other (the distance matrix): lengths apart by up to ~60
            mot = rand_idx(rng, int(rng.integers(2, 7)))
            L = int(rng.integers(8, 100))
            a = np.tile(mot, L // len(mot) + 1)[:L]
            L2 = max(0, L + int(rng.integers(-60, 61)))
            b = mutate_idx(rng, np.tile(mot, L2 // len(mot) + 1)[:L2], 0.01, 0.005, 0.005)
            if i % 9 == 0:
                b = rand_idx(rng, int(rng.integers(0, 101)))
        elif kind == "short":      # <= 100 bp (edit-distance matrix of the cluster genotyper)
            a = rand_idx(rng, int(rng.integers(0, 101)))
            b = mutate_idx(rng, a, 0.05, 0.03, 0.03) if i % 4 else rand_idx(rng, int(rng.integers(0, 101)))
        else:                      # generic
            a = rand_idx(rng, int(rng.integers(0, 320)))
            if i % 5 == 0:
                b = rand_idx(rng, int(rng.integers(0, 150)))
            else:
                b = mutate_idx(rng, a, 0.03, 0.02, 0.02)
                if kind == "embedded":
                    b = np.concatenate([rand_idx(rng, int(rng.integers(0, 150))), b, rand_idx(rng, int(rng.integers(0, 150)))])
            if i % 11 == 0:
                a, b = b, a
        pats.append(tobytes(a))
        txts.append(tobytes(b))
    return pats, txts


def compare(got, ref, coff, n):
    bad = []
    no_ops = got["ops"] is None
    for f in ("status", "score", "n_match", "span4", "cigar_len") + (() if no_ops else ("ops_len",)):
        g, r = np.asarray(got[f]), np.asarray(ref[f])
        if not np.array_equal(g, r):
            d = np.nonzero((g != r).reshape(n, -1).any(axis=1))[0]
            bad.append((f, int(d[0]), len(d)))
    if not bad:
        for j in range(n):
            o, cl, ol = int(coff[j]), int(ref["cigar_len"][j]), int(ref["ops_len"][j])
            if not np.array_equal(got["cigar"][o:o + cl], ref["cigar"][o:o + cl]) or (not no_ops and bytes(got["ops"][o:o + ol]) != bytes(ref["ops"][o:o + ol])):
                bad.append(("cigar/ops", j, 1))
                break
    return bad


def run_mode(name, al, span, free, op, pats, txts, threads, min_length=None, want_ops=True):
    n = len(pats)
    p = al._params(span, *free)
    if min_length is not None:
        p.bialign_min_length = min_length
    t0 = time.perf_counter()
    got = al._run_batch(p, pats, txts, want_ops)
    tg = time.perf_counter() - t0
    blob = b"".join(pats) + b"".join(txts)
    plen = np.array([len(x) for x in pats], np.uint32)
    tlen = np.array([len(x) for x in txts], np.uint32)
    pat_off = np.zeros(n, np.uint64); pat_off[1:] = np.cumsum(plen[:-1], dtype=np.uint64)
    txt_off = np.zeros(n, np.uint64); txt_off[1:] = np.cumsum(tlen[:-1], dtype=np.uint64)
    txt_off += np.uint64(int(plen.sum()))
    coff = got["cigar_off"]
    batch = dict(seqs=np.frombuffer(blob, np.uint8).copy(), pat_off=pat_off, pat_len=plen, txt_off=txt_off, txt_len=tlen, cigar_off=coff, ops_off=coff)
    t0 = time.perf_counter()
    ref = oracle.wfa_batch(op, batch, n_threads=threads)
    tc = time.perf_counter() - t0
    bad = compare(got, ref, coff, n)
    st = {int(a): int(b) for a, b in zip(*np.unique(ref["status"], return_counts=True))}
    print("[wfa fuzz] %-52s jobs %6d  gpu %.2fs oracle %.1fs  statuses %s  %s" % (name, n, tg, tc, st, "OK" if not bad else "MISMATCH %s" % bad), flush=True)
    return len(bad)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    threads = min(os.cpu_count() or 1, 128)
    A, S, M, H = W.AlignmentScope, W.MemoryModel, None, W.Heuristic
    nbad = nmodes = njobs = 0
    k = 0

    def rng():
        nonlocal k
        k += 1
        return np.random.default_rng(seed * 1000 + k)

    only = os.environ.get("WFA_FUZZ_ONLY", "")  # e.g. "no ops": only the modes whose name contains it

    def go(name, al, span, free, op, kind, count=n, **kw):
        nonlocal nbad, nmodes, njobs
        if only and only not in name:
            return
        pats, txts = gen_pairs(rng(), count, kind)
        if span == "endsfree" and free[0] >= 0 and max(free) > 0:   # fixed free-end lengths must not exceed the sequences
            r = rng()
            pats = [x + tobytes(rand_idx(r, max(free[0], free[1]))) if len(x) < max(free[0], free[1]) else x for x in pats]
            txts = [x + tobytes(rand_idx(r, max(free[2], free[3]))) if len(x) < max(free[2], free[3]) else x for x in txts]
        nbad += run_mode(name, al, span, free, op, pats, txts, threads, **kw)
        nmodes += 1
        njobs += count

    # flank configuration (dedicated kernel, compile-time specialisation) and other penalties / shapes on the same kernel
    hi = lambda b: b.with_heuristic(H.none()).build()
    go("affine(2,5,1) text-free ends, flank-like", hi(W.WFAligner.builder(A.Alignment, S.MemoryHigh).affine(2, 5, 1)), "endsfree", (0, 0, -1, -1),
       oracle.wfa_params(metric="affine", x=2, o1=5, e1=1, span="endsfree", pbf=0, pef=0, tbf=-1, tef=-1, heuristic="none"), "flank")
    for (x, o, e) in ((1, 0, 1), (3, 2, 2), (4, 6, 2), (7, 9, 1), (2, 5, 1)):
        al = hi(W.WFAligner.builder(A.Alignment, S.MemoryHigh).affine(x, o, e))
        go("affine(%d,%d,%d) end-to-end, exact" % (x, o, e), al, "end2end", (0, 0, 0, 0), oracle.wfa_params(metric="affine", x=x, o1=o, e1=e, heuristic="none"), "generic", n // 2)
        go("affine(%d,%d,%d) text-free ends" % (x, o, e), al, "endsfree", (0, 0, -1, -1),
           oracle.wfa_params(metric="affine", x=x, o1=o, e1=e, span="endsfree", pbf=0, pef=0, tbf=-1, tef=-1, heuristic="none"), "embedded", n // 2)
        go("affine(%d,%d,%d) ends-free (3,5,40,7)" % (x, o, e), al, "endsfree", (3, 5, 40, 7),
           oracle.wfa_params(metric="affine", x=x, o1=o, e1=e, span="endsfree", pbf=3, pef=5, tbf=40, tef=7, heuristic="none"), "embedded", n // 2)
    # every metric, end-to-end, exact and with the default heuristic; score-only too
    mk = {"indel": lambda b: b.indel(), "edit": lambda b: b.edit(), "linear": lambda b: b.linear(6, 2), "affine": lambda b: b.affine(6, 4, 2),
          "affine2p": lambda b: b.affine2p(8, 4, 2, 24, 1)}
    pen = {"indel": {}, "edit": {}, "linear": dict(x=6, e1=2), "affine": dict(x=6, o1=4, e1=2), "affine2p": dict(x=8, o1=4, e1=2, o2=24, e2=1)}
    for metric in mk:
        for heur in ("none", "default"):
            b = mk[metric](W.WFAligner.builder(A.Alignment, S.MemoryHigh))
            al = b.with_heuristic(H.none()).build() if heur == "none" else b.build()
            go("%s end-to-end, heuristic %s" % (metric, heur), al, "end2end", (0, 0, 0, 0), oracle.wfa_params(metric=metric, heuristic=heur, **pen[metric]), "generic", n // 2)
        al = mk[metric](W.WFAligner.builder(A.Score, S.MemoryHigh)).with_heuristic(H.none()).build()
        go("%s end-to-end, score only" % metric, al, "end2end", (0, 0, 0, 0), oracle.wfa_params(metric=metric, scope="score", heuristic="none", **pen[metric]), "generic", n // 2)
    # memory modes
    for mem, mm in (("med", S.MemoryMed), ("low", S.MemoryLow)):
        al = hi(W.WFAligner.builder(A.Alignment, mm).affine(2, 5, 1))
        go("affine(2,5,1) end-to-end, memory %s" % mem, al, "end2end", (0, 0, 0, 0), oracle.wfa_params(metric="affine", x=2, o1=5, e1=1, memory=mem, heuristic="none"), "generic", n // 2)
    # BiWFA: consensus configuration (default heuristic), both readings of the short-sequence fallback; exact; edit score-only
    for ml in (100, 0):
        al = W.WFAligner.builder(A.Alignment, S.MemoryUltraLow).affine(2, 5, 1).build()
        go("BiWFA affine(2,5,1) consensus-like, min_length %d" % ml, al, "end2end", (0, 0, 0, 0),
           oracle.wfa_params(metric="affine", x=2, o1=5, e1=1, memory="ultralow", heuristic="default", min_length=ml), "str", min_length=ml)
    al = hi(W.WFAligner.builder(A.Alignment, S.MemoryUltraLow).affine(2, 5, 1))
    go("BiWFA affine(2,5,1) exact", al, "end2end", (0, 0, 0, 0), oracle.wfa_params(metric="affine", x=2, o1=5, e1=1, memory="ultralow", heuristic="none"), "generic", n // 2)
    al = W.WFAligner.builder(A.Score, S.MemoryUltraLow).edit().build()
    go("BiWFA edit score-only (get_dist)", al, "end2end", (0, 0, 0, 0), oracle.wfa_params(metric="edit", scope="score", memory="ultralow", heuristic="default"), "short", 2 * n)
    go("BiWFA edit score-only, longer", al, "end2end", (0, 0, 0, 0), oracle.wfa_params(metric="edit", scope="score", memory="ultralow", heuristic="default"), "str", n // 2)
    # the same BiWFA configurations WITHOUT expanded operations -- what the locus path asks for, and the form the register-resident kernel
    # (wfa_lean.hip) takes in front of the generic one -- on pairs shaped like its callers'
    for ml in (100, 0):
        for heur in ("default", "none"):
            b = W.WFAligner.builder(A.Alignment, S.MemoryUltraLow).affine(2, 5, 1)
            al = b.build() if heur == "default" else hi(b)
            for kind in ("consensus", "str", "alleles", "generic"):
                go("BiWFA affine(2,5,1) no ops, %s, heuristic %s, min_length %d" % (kind, heur, ml), al, "end2end", (0, 0, 0, 0),
                   oracle.wfa_params(metric="affine", x=2, o1=5, e1=1, memory="ultralow", heuristic=heur, min_length=ml), kind, n // 2, min_length=ml, want_ops=False)
    for heur in ("default", "none"):
        b = W.WFAligner.builder(A.Score, S.MemoryUltraLow).edit()
        al = b.build() if heur == "default" else hi(b)
        for kind in ("short", "alleles", "consensus", "str"):
            go("BiWFA edit score-only no ops, %s, heuristic %s" % (kind, heur), al, "end2end", (0, 0, 0, 0),
               oracle.wfa_params(metric="edit", scope="score", memory="ultralow", heuristic=heur), kind, n, want_ops=False)
        b = W.WFAligner.builder(A.Alignment, S.MemoryUltraLow).edit()
        al = b.build() if heur == "default" else hi(b)
        go("BiWFA edit alignment no ops, consensus, heuristic %s" % heur, al, "end2end", (0, 0, 0, 0),
           oracle.wfa_params(metric="edit", memory="ultralow", heuristic=heur), "consensus", n // 2, want_ops=False)
    print("RESULT wfa fuzz: modes=%d jobs=%d modes_with_mismatch=%d seed=%d" % (nmodes, njobs, nbad, seed))
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
