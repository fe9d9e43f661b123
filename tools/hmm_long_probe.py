"""One or a few very long alleles through trgt_hmm_batch (the tail of a cfg3 call): wall time; with `make HMMPROF=1` the phase split."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from trgt_amd import hmm
rng = np.random.default_rng(7)
B = np.frombuffer(b"ACGT", np.uint8)
def seq(motif, L):
    s = (motif * (L // len(motif) + 1))[:L]
    a = np.frombuffer(s, np.uint8).copy(); m = rng.random(L) < 0.01; a[m] = B[rng.integers(0, 4, int(m.sum()))]
    return a.tobytes()
for name, motifs, L, n in (("CAG x 10 kb, 4 jobs", [b"CAG"], 10000, 4), ("CAG x 10 kb, 1 job", [b"CAG"], 10000, 1), ("CAG x 2.5 kb, 4 jobs", [b"CAG"], 2500, 4),
                           ("GGCCCC 10 kb 2 jobs", [b"GGCCCC"], 10000, 2), ("20-mer 10 kb 2 jobs", [b"ACGTTGCAAGGCTTAACCGT"], 10000, 2)):
    b = hmm.pack_hmm_batch([motifs], [(0, seq(motifs[0], L)) for _ in range(n)])
    for rep in range(3):
        t0 = time.perf_counter()
        hmm.hmm_batch(b, want_path=False)
        dt = time.perf_counter() - t0
    print("%-24s %.2f ms" % (name, dt * 1e3), flush=True)
