import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trgt_amd import _lib, locus, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
ctx = _lib.Context(0)
batches = [synth.generate(300, first_locus=7000 + 1000 * i, config=2) for i in range(4)]
ref = [locus.run_batch(b, ctx=ctx) for b in batches]
print("cfg2 blocking ok", flush=True)
if mode in ("full", "submit"):
    pins = [torch.from_numpy(b["read_blob"]).pin_memory() for b in batches]
    t = locus.submit_batch(batches[0], ctx=ctx, reads=pins[0])
    t2 = locus.submit_batch(batches[1], ctx=ctx, reads=pins[1])
    t.wait(); t2.wait()
    print("cfg2 submit/wait ok", flush=True)
if mode in ("full", "close"):
    ctx.close()
    ctx = _lib.Context(0)
    print("new ctx", flush=True)
for i in range(4):
    b = synth.generate(60, first_locus=7000 + 1000 * i, config=5)
    out = locus.run_batch(b, ctx=ctx)
    print("  cfg5 batch", i, "ok", int(out.stats[1]), "consensus jobs", flush=True)
