"""HMM kernel alone: wall time per batch for a few (motif set, sequence length) shapes; with `make HMMPROF=1` the phase split is printed.
usage: python tools/hmm_probe.py [reps]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from trgt_amd import hmm  # noqa: E402

rng = np.random.default_rng(7)
B = np.frombuffer(b"ACGT", np.uint8)


def rnd(n):
    return B[rng.integers(0, 4, n)].tobytes()


def jobs_for(motifs, n, lo, hi):
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        s = bytearray()
        while len(s) < L:
            s += motifs[int(rng.integers(0, len(motifs)))]
        a = np.frombuffer(bytes(s[:L]), np.uint8).copy()
        m = rng.random(L) < 0.01
        a[m] = B[rng.integers(0, 4, int(m.sum()))]
        out.append(a.tobytes())
    return out


shapes = [("1 motif of 3 bp, 2000 x 60-600 bp", [rnd(3)], 2000, 60, 600),
          ("2 motifs of 5 bp, 2000 x 60-600 bp", [rnd(5), rnd(5)], 2000, 60, 600),
          ("1 motif of 20 bp, 2000 x 60-600 bp", [rnd(20)], 2000, 60, 600),
          ("3 motifs 3/5/6 bp, 140 x 500-10000 bp", [rnd(3), rnd(5), rnd(6)], 140, 500, 10000),
          ("1 motif of 60 bp, 140 x 500-10000 bp", [rnd(60)], 140, 500, 10000)]


def many_sets(n_sets, n_jobs, lo_copies, hi_copies):
    sets = [[rnd(int(rng.integers(2, 7)))] for _ in range(n_sets)]
    jobs = []
    for j in range(n_jobs):
        si = j % n_sets
        m = sets[si][0]
        a = np.frombuffer(m * int(rng.integers(lo_copies, hi_copies + 1)), np.uint8).copy()
        e = rng.random(len(a)) < 0.01
        a[e] = B[rng.integers(0, 4, int(e.sum()))]
        jobs.append((si, a.tobytes()))
    return sets, jobs


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, motifs, n, lo, hi in shapes:
    seqs = jobs_for(motifs, n, lo, hi)
    batch = hmm.pack_hmm_batch([motifs], [(0, s) for s in seqs])
    hmm.hmm_batch(batch)
    t0 = time.perf_counter()
    for _ in range(reps):
        hmm.hmm_batch(batch)
    dt = (time.perf_counter() - t0) / reps
    cols = sum(len(s) for s in seqs)
    print("%-44s states %4d  %8.3f ms/batch  %7.1f ns/column (all jobs in parallel)" % (name, hmm.num_states(motifs), dt * 1e3, dt * 1e9 / cols))

for name, (sets, jobs) in [("whole-genome STRs: 20000 x (2-6 bp x 10-40)", many_sets(1000, 20000, 10, 40)),
                           ("longer STR alleles: 20000 x (2-6 bp x 10-300)", many_sets(1000, 20000, 10, 300))]:
    batch = hmm.pack_hmm_batch(sets, jobs)
    hmm.hmm_batch(batch, want_path=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        hmm.hmm_batch(batch, want_path=False)
    dt = (time.perf_counter() - t0) / reps
    cols = sum(len(s) for _, s in jobs)
    print("%-48s %8.3f ms/batch (host call)  %7.3f ns/column" % (name, dt * 1e3, dt * 1e9 / cols))
