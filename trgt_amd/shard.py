"""Locus sharding across GPUs: loci are independent (the reference runs one rayon task per locus,
src/commands/genotype.rs:179-187), so rank r of N owns the contiguous locus range [r*n/N, (r+1)*n/N) and there is no
data-path collective; only the timing (max over ranks) and, optionally, a result digest are reduced."""
import hashlib

import numpy as np


def shard_range(rank, world, n_loci):
    """Contiguous, exhaustive, non-overlapping split of range(n_loci)."""
    lo = (rank * n_loci) // world
    hi = ((rank + 1) * n_loci) // world
    return lo, hi


def result_digest(outputs, n_loci):
    """Order-sensitive digest of a BatchOutputs-like object: equal digests <=> byte-identical allele calls/annotations."""
    h = hashlib.sha256()
    for l in range(n_loci):
        for a in range(int(outputs.n_alleles[l])):
            s = 2 * l + a
            o, n = int(outputs.allele_off[s]), int(outputs.allele_len[s])
            h.update(bytes(outputs.allele_blob[o:o + n]))
            so, ns = int(outputs.span_off[s]), int(outputs.n_spans[s])
            h.update(np.ascontiguousarray(outputs.spans3[3 * so:3 * (so + ns)]).tobytes())
            h.update(np.float64(outputs.purity[s]).tobytes())
        h.update(np.ascontiguousarray(outputs.ci[4 * l:4 * l + 4]).tobytes())
    return h.hexdigest()


def max_over_ranks(value, dist=None, device=None):
    """MAX all-reduce of a python float (nccl on GPU, gloo on CPU); identity when not distributed."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
