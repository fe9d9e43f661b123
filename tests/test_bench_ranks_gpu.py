"""bench.py with more than one rank: the launch the driver uses for the scaling runs (torch.distributed.run, one process per GPU), here
with both ranks on the one GPU of the test box (TRGT_BENCH_ONE_GPU=1: gloo instead of RCCL, every rank on device 0).  Everything else is
the N > 1 path as it stands: disjoint locus ranges per rank, the barrier-bracketed timed region, MAX over ranks, one JSON line from
rank 0, and every rank's recomputation of its neighbour's shard with the digests compared."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_print_one_line_and_agree_on_each_others_shards():
    env = dict(os.environ, TRGT_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--contexts", "2",
           "--loci", "1500", "--no-cpu-baseline", "--no-streaming"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["metric"] == "loci/s" and d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["config"]["loci_per_gpu"] == 1500
    assert d["multi_gpu_digest_check"] == {"ranks": 2, "shards_recomputed_on_another_gpu": 2, "digest_mismatches": 0}
    # value is the whole job: both ranks' loci over the slower rank's time
    assert abs(d["value"] - 2 * 1500 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02
