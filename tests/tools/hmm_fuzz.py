"""Large seeded fuzz of trgt_hmm_batch against the CPU oracle: random motif sets (1-10 motifs of 1-30 bp, some with N), alleles of
0-1500 bp made of motif runs with interruptions and sequencing-like errors, non-ACGT bytes, a few 10-kb alleles.  Compares state
paths, spans, motif counts, edit / max distances and the bits of the f64 purity of every job.  A developer tool (the GPU suite keeps
small batches of the same shapes: tests/test_hmm_gpu.py).

    python tests/tools/hmm_fuzz.py [sets_per_round=2000] [rounds=10] [seed=1]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from trgt_amd import hmm as H
from oracle import binding as oracle

LUT = np.frombuffer(b"ACGT", np.uint8)


def rand_seq(rng, n):
    return LUT[rng.integers(0, 4, n)].tobytes()


def motif(rng, lo, hi):
    m = bytearray(rand_seq(rng, int(rng.integers(lo, hi + 1))))
    if rng.random() < 0.15:
        m[int(rng.integers(0, len(m)))] = ord("N")
    return bytes(m)


def allele(rng, motifs, total, err):
    parts, n = [], 0
    while n < total:
        m = motifs[int(rng.integers(0, len(motifs)))]
        c = int(rng.integers(1, 14))
        unit = np.frombuffer(m, np.uint8)
        run = np.tile(unit, c)
        nn = run == ord("N")
        if nn.any():
            run = run.copy(); run[nn] = LUT[rng.integers(0, 4, int(nn.sum()))]
        parts.append(run); n += len(run)
        if rng.random() < 0.2:
            k = int(rng.integers(1, 9)); parts.append(np.frombuffer(rand_seq(rng, k), np.uint8)); n += k
    a = np.concatenate(parts)[:total] if parts else np.zeros(0, np.uint8)
    if len(a) and err > 0:
        a = a[rng.random(len(a)) >= err / 2]
        s = rng.random(len(a)) < err
        a = np.where(s, LUT[rng.integers(0, 4, len(a))], a).astype(np.uint8)
        k = (rng.random(len(a)) < err / 2).astype(np.int64)
        out = np.repeat(a, 1 + k)
        at = (np.cumsum(1 + k) - 1)[k > 0]
        out[at] = LUT[rng.integers(0, 4, len(at))]
        a = out
    return a.tobytes()


def compare(batch, got, ref, n_jobs, want_path):
    for f in ("n_spans", "path_len", "edit", "maxd", "counts"):
        if not np.array_equal(got[f], ref[f]):
            return f
    if not np.array_equal(got["purity"].view(np.uint64), ref["purity"].view(np.uint64)):
        return "purity bits"
    for j in range(n_jobs):
        so, ns = int(batch["span_off"][j]), int(ref["n_spans"][j])
        if not np.array_equal(got["spans"][3 * so:3 * (so + ns)], ref["spans"][3 * so:3 * (so + ns)]):
            return "spans of job %d" % j
        if want_path:
            po, pl = int(batch["path_off"][j]), int(ref["path_len"][j])
            if not np.array_equal(got["path"][po:po + pl], ref["path"][po:po + pl]):
                return "path of job %d" % j
    return None


def main():
    n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    threads = min(os.cpu_count() or 1, 128)
    bad = jobs_total = 0
    cells = 0
    for r in range(rounds):
        rng = np.random.default_rng(seed * 100 + r)
        kind = ("str", "multi", "long motifs", "mixed lengths", "dirty bytes")[r % 5]
        sets, jobs = [], []
        for s in range(n_sets):
            if kind == "str":
                ms = [motif(rng, 2, 6)]
            elif kind == "long motifs":
                # (one VNTR motif of up to 62 bases: 32- and 64-lane jobs of the position-per-lane fill whose deletion chain goes row by
                #  row, hmm_ppl.hpp; or a few shorter ones: often more positions than a wave has lanes, the older fill)
                ms = [motif(rng, 7, 62)] if rng.random() < 0.5 else [motif(rng, 7, 30) for _ in range(int(rng.integers(1, 4)))]
            else:
                ms = [motif(rng, 1, 12) for _ in range(int(rng.integers(1, 11)))]
            sets.append(ms)
            for _ in range(2):
                total = int(rng.integers(0, 250)) if kind != "mixed lengths" else int(rng.choice([0, 1, 2, 5, 40, 300, 900, 1500]))
                a = allele(rng, ms, total, float(rng.choice([0.0, 0.01, 0.03, 0.1])))
                if kind == "dirty bytes" and len(a) > 3:
                    b = bytearray(a)
                    for _ in range(int(rng.integers(1, 4))):
                        b[int(rng.integers(0, len(b)))] = int(rng.choice(list(b"NRYacgtn-")))
                    a = bytes(b)
                jobs.append((s, a))
        if r == rounds - 1:   # a few very long alleles (cfg3-like)
            for s in range(4):
                jobs.append((s, allele(rng, sets[s], int(rng.integers(5000, 10001)), 0.005)))
        want_path = r % 2 == 0
        batch = H.pack_hmm_batch(sets, jobs)
        t0 = time.perf_counter()
        got = H.hmm_batch(batch, want_path=want_path)
        tg = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref = oracle.hmm_batch(batch, n_threads=threads, want_path=want_path)
        tc = time.perf_counter() - t0
        why = compare(batch, got, ref, len(jobs), want_path)
        bad += why is not None
        jobs_total += len(jobs)
        print("[hmm fuzz] round %2d %-14s sets %5d jobs %6d paths %-3s gpu %.2fs oracle %.1fs  %s" % (r, kind, n_sets, len(jobs), "yes" if want_path else "no", tg, tc, "OK" if why is None else "MISMATCH: " + why), flush=True)
    print("RESULT hmm fuzz: rounds=%d jobs=%d rounds_with_mismatch=%d seed=%d" % (rounds, jobs_total, bad, seed))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
