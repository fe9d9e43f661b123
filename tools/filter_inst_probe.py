"""Time of the pre-filter on alignments that hold no flank (the expensive kind) for a given text length, per kernel instantiation
(TRGT_FILTER_FORCE=42 / 52 / 71 / 91 picks wfa_filter_kernel<4,2> / <5,2> / <7,1> / <9,1> for the whole launch)."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
tl = int(sys.argv[1]) if len(sys.argv) > 1 else 640
from trgt_amd import wfaligner, _lib
rng = np.random.default_rng(3)
B = np.frombuffer(b"ACGT", np.uint8)
n = 12000
pats = [B[rng.integers(0, 4, 250)].tobytes() for _ in range(n)]
txts = [B[rng.integers(0, 4, tl)].tobytes() for _ in range(n)]
ctx = _lib.Context(0)
r = None
for rep in range(3):
    t0 = time.perf_counter()
    r = wfaligner.flank_filter_batch(pats, txts, 175, ctx=ctx, early_reject=True)
    dt = time.perf_counter() - t0
print("force=%s tlen=%d: %.2f ms, kept %d, offsets %.3g, checksum %d" % (os.environ.get("TRGT_FILTER_FORCE", "-"), tl, dt * 1e3, int(r["keep"].sum()), r["offsets"], int(r["bound"].astype(np.int64).sum())))
