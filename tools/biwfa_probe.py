"""Time of the generic kernel on consensus-like BiWFA batches: n pairs of `length` bases, identical / one substitution / one deleted base."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trgt_amd import _lib, wfaligner as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(1)
B = np.frombuffer(b"ACGT", np.uint8)
base = [B[rng.integers(0, 4, L)].tobytes() for _ in range(n)]
def sub(s):
    b = bytearray(s); p = L // 2; b[p] = ord("A") if b[p] != ord("A") else ord("C"); return bytes(b)
ctx = _lib.Context(0)
if len(sys.argv) > 3:  # workspace limit in GB: fewer resident workgroups (ws_limit / workspace per workgroup)
    ctx.check(_lib.lib().trgt_hip_set_workspace_limit(ctx.handle, int(float(sys.argv[3]) * (1 << 30))))
al = W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryUltraLow).affine(2, 5, 1).build(ctx)
for name, txt in (("identical", base), ("one substitution", [sub(s) for s in base]), ("one deleted base", [s[:L // 3] + s[L // 3 + 1:] for s in base])):
    for rep in range(2):
        t0 = time.perf_counter()
        r = al.align_end_to_end_batch(base, txt, want_ops=False)
        dt = time.perf_counter() - t0
    ctx.timing_enable(True); ctx.timing_reset()
    al.align_end_to_end_batch(base, txt, want_ops=False)
    k = ctx.timing_get(1)
    ctx.timing_enable(False)
    print("%-18s %d pairs of %d: call %.2f ms, kernel %.3f ms" % (name, n, L, dt * 1e3, k[0]))
