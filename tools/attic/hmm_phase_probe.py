import sys, numpy as np
sys.path.insert(0, ".")
from trgt_amd import hmm, _lib
import os
for motifs in ([b"CAG"], [b"AAGGG", b"AAAAG", b"ACG"], [b"ACGTACGTACGTACGTAC"]):
    seq = (motifs[0] * 4000)[:10000]
    batch = hmm.pack_hmm_batch([motifs], [(0, seq)])
    for env in ({}, {"TRGT_HMM_FOUR_ROUNDS": 1}):
        ctx = _lib.context_with_env(**env)
        hmm.hmm_batch(batch, ctx=ctx)
        print("==", motifs, env, "states", hmm.num_states(motifs), file=sys.stderr, flush=True)
        hmm.hmm_batch(batch, ctx=ctx)
        ctx.close()
