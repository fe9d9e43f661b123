"""world_size-2 run of the N>1 path on CPU (gloo): every rank generates and processes its own contiguous locus
shard with no data-path collective; only the max-over-ranks time and a digest are reduced.  The per-shard compute
here is the CPU oracle (no GPU in this container) -- what is under test is the sharding / aggregation logic that
bench.py uses, and that sharded results concatenate to the unsharded ones."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _analyze(orc, b, l):
    a0, a1 = int(b["locus_read_begin"][l]), int(b["locus_read_begin"][l + 1])
    reads = [bytes(b["read_blob"][int(b["read_off"][r]):int(b["read_off"][r]) + int(b["read_len"][r])]) for r in range(a0, a1)]
    lf = bytes(b["flank_blob"][int(b["lf_off"][l]):int(b["lf_off"][l]) + 250])
    rf = bytes(b["flank_blob"][int(b["rf_off"][l]):int(b["rf_off"][l]) + 250])
    tr = bytes(b["tr_blob"][int(b["tr_off"][l]):int(b["tr_off"][l]) + int(b["tr_len"][l])])
    motifs = [bytes(b["motif_blob"][int(b["motif_off"][l]):int(b["motif_off"][l + 1])])]
    r = orc.locus_analyze(lf, rf, tr, motifs, reads)
    return (tuple(r["alleles"]), r["MC"], r["MS"], r["AP"], r["ALLR"], r["SD"])


def _worker(rank, world, port, n_loci, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import binding as orc
    from trgt_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(rank, world, n_loci)
    b = synth.generate(hi - lo, first_locus=lo)
    res = [_analyze(orc, b, l) for l in range(hi - lo)]
    slowest = shard.max_over_ranks(1.0 + rank, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, res))
    if rank == 0:
        q.put((slowest, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process(oracle):
    import torch.multiprocessing as mp
    from trgt_amd import shard, synth
    n_loci, world = 7, 2
    assert [shard.shard_range(r, world, n_loci) for r in range(world)] == [(0, 3), (3, 7)]
    assert [shard.shard_range(r, 8, 10**6)[1] - shard.shard_range(r, 8, 10**6)[0] for r in range(8)] == [125000] * 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_loci, q)) for r in range(world)]
    for p in procs:
        p.start()
    slowest, gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert slowest == 2.0  # max over ranks
    whole = synth.generate(n_loci)
    want = [_analyze(oracle, whole, l) for l in range(n_loci)]
    got = [None] * n_loci
    for lo, hi, res in gathered:
        got[lo:hi] = res
    assert got == want
