import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from trgt_amd import wfaligner as W, _lib
from helpers import rand_dna, mutate, repeat_allele
rng = np.random.default_rng(5)
pats, txts = [], []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 7):
    a = repeat_allele(rng, [b"GCN", b"GCA"], int(rng.integers(100, 200)), err=0.0) if i % 2 else rand_dna(rng, int(rng.integers(100, 200)))
    pats.append(bytes(a)); txts.append(bytes(mutate(rng, a, 0.01, 0.01, 0.01)))
pats.append(bytes(rand_dna(rng, 256))); txts.append(bytes(rand_dna(rng, 256)))
al = W.WFAligner.builder(W.AlignmentScope.Alignment, W.MemoryModel.MemoryUltraLow).affine(2, 5, 1).build()
for rep in range(3):
    r = al.align_end_to_end_batch(pats, txts, want_ops=False)
    print("rep", rep, "status", list(r["status"]), "score", list(r["score"]), flush=True)
