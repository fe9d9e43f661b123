#!/usr/bin/env python3
"""bench.py -- loci/s of the TRGT hot path on MI355X (BASELINE.json metric), one process per GPU.

A "step" = one full pass of trgt_locus_batch over one batch of synthetic loci per GPU: flank location (exact scan, register-
resident pre-filter + back-tracing ends-free WFA for the misses) -> length / cluster genotyping -> consensus BiWFA where
needed -> motif-HMM labelling (MS / MC / AP).

  python bench.py [--gpus 1] [--steps 100] [--warmup 3] [--config 2|3|4|5] [--loci N]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

--config picks the BASELINE.json workload: 2 = configs[1] (10k single-motif STR loci, the configuration the metric is quoted on),
3 = configs[2] (70 loci of the pathogenic catalog, alleles to 10 kb), 4 = configs[3] catalog mix (10k loci per GPU per step),
5 = configs[4] (compound / N motifs through the cluster genotyper).

`value` is measured with read and flank bytes resident in HBM before the timed region (only offsets / spans / alleles cross PCIe
inside it); `value_streaming` is the same batches with the reads starting in pinned HOST memory, uploaded inside every call.
Loci shard embarrassingly: rank r generates and processes loci [r*L, (r+1)*L) (weak scaling, no collective on the data path);
value = loci of all ranks * steps / max-over-ranks time.  At N > 1 every rank also recomputes its right neighbour's shard
(untimed) and the digests must agree: N-GPU output == 1-GPU output, byte for byte.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md; ~6.3 TB/s achievable, measured below too)
VALU_LANES_PER_CU = 64   # 4 SIMDs x 16 lanes: one wave64 VALU instruction occupies a SIMD for 4 cycles (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles)
CLOCK_GHZ = 2.4

DEFAULT_LOCI = {2: 10000, 3: 70, 4: 10000, 5: 2000}
WORKLOAD = {
    2: "BASELINE configs[1]: %d synthetic single-motif STR loci per GPU (motif 3-6 bp, allele <= 200 bp), 30 reads/locus HiFi-like, 10%% truncated reads (SURVEY.md Appendix E), seed 20250509",
    3: "BASELINE configs[2]: %d loci per GPU over the 56 motif sets of the pathogenic catalog, one allele of 10-40 units and one expanded allele log-uniform in 500 bp .. 10 kb, 30 reads/locus, 10%% truncated (trgt_amd.synth.generate_cfg3)",
    4: "BASELINE configs[3] per-GPU shard: %d loci of the genome-wide catalog mix (70%% STR, 20%% 2-5 motifs, 10%% VNTR motifs of 7-60 bp), 30 reads/locus, 10%% truncated",
    5: "BASELINE configs[4]: %d compound / N-motif loci per GPU through the cluster genotyper, 30 reads/locus, 10%% truncated",
}


def make_batch(config, n, first):
    from trgt_amd import synth
    if config == 3:
        return synth.generate_cfg3(n, first_locus=first)
    return synth.generate(n, first_locus=first, config=config)


def locus_inputs(batch, l):
    a0, a1 = int(batch["locus_read_begin"][l]), int(batch["locus_read_begin"][l + 1])
    reads = [bytes(batch["read_blob"][int(batch["read_off"][r]):int(batch["read_off"][r]) + int(batch["read_len"][r])]) for r in range(a0, a1)]
    lf = bytes(batch["flank_blob"][int(batch["lf_off"][l]):int(batch["lf_off"][l]) + int(batch["lf_len"][l])])
    rf = bytes(batch["flank_blob"][int(batch["rf_off"][l]):int(batch["rf_off"][l]) + int(batch["rf_len"][l])])
    tr = bytes(batch["tr_blob"][int(batch["tr_off"][l]):int(batch["tr_off"][l]) + int(batch["tr_len"][l])])
    m0, m1 = int(batch["set_motif_begin"][l]), int(batch["set_motif_begin"][l + 1])
    motifs = [bytes(batch["motif_blob"][int(batch["motif_off"][m]):int(batch["motif_off"][m + 1])]) for m in range(m0, m1)]
    return lf, rf, tr, motifs, reads, (a0, a1)


def cpu_baseline(batch, out, config, seconds_budget=20.0, max_loci=8000):
    """The CPU oracle (port of the reference algorithms, single thread) timed on a bounded sample of the same batch; what it
    computed is then compared with what the GPU path returned for the same loci (outside the timed region)."""
    from oracle import binding as orc
    from trgt_amd import locus
    orc.lib()
    n = min(int(batch["n_loci"]), max_loci)
    genotyper = batch.get("genotyper")
    refs = []
    t_cpu = 0.0
    for l in range(n):
        lf, rf, tr, motifs, reads, _ = locus_inputs(batch, l)
        kw = {"genotyper": 1} if genotyper is not None and int(genotyper[l]) == 1 else {}
        kw["ploidy"] = int(batch["ploidy"][l])
        t0 = time.perf_counter()
        refs.append(orc.locus_analyze(lf, rf, tr, motifs, reads, **kw))
        t_cpu += time.perf_counter() - t0
        if t_cpu > seconds_budget:
            break
    done = len(refs)
    mismatches = 0
    for l, ref in enumerate(refs):
        a0, a1 = int(batch["locus_read_begin"][l]), int(batch["locus_read_begin"][l + 1])
        got = locus.locus_result(batch, out, l)
        f = got.vcf_fields()
        ok = (np.array_equal(out.span_start[a0:a1], ref["span_start"]) and np.array_equal(out.span_end[a0:a1], ref["span_end"]) and
              [a.seq.decode() for a in got.genotype] == ref["alleles"] and all(f[k] == ref[k] for k in ("AL", "ALLR", "SD", "MC", "MS", "AP")))
        mismatches += 0 if ok else 1
    return (dict(value=round(done / t_cpu, 2), unit="loci/s", cores=1, kind="port",
                 sample="first %d loci of the same synthetic batch, oracle/liboracle.so (C++ restatement of the reference algorithms), 1 thread, %.1f s" % (done, t_cpu)),
            dict(parity_checked_loci=done, mismatches=mismatches,
                 compared="per-read spans, allele sequences, AL / ALLR / SD / MC / MS / AP of the last timed step vs the oracle"))


def cpu_baseline_mt(batch, threads):
    """The same oracle on `threads` native host threads (orc_locus_analyze_many, dynamic chunks of loci: the reference runs one
    rayon task per locus).  Reported next to the single-thread figure; it is still the C++ restatement, not the reference binary."""
    from oracle import binding as orc
    orc.lib()
    n = int(batch["n_loci"])
    orc.locus_analyze_many(batch, 0, min(n, 4 * threads), threads)  # thread start-up, page faults
    t0 = time.perf_counter()
    done, _ = orc.locus_analyze_many(batch, 0, n, threads)
    dt = time.perf_counter() - t0
    return dict(value=round(done / dt, 2), unit="loci/s", cores=threads, kind="port", cpu_quota=cpu_quota(),  # (threads share the quota when there is one)
                sample="all %d loci of the same synthetic batch, oracle/liboracle.so, %d native threads pulling chunks of loci from a shared counter, %.1f s" % (done, threads, dt))


def copy_peak_gbs(torch, nbytes=1 << 30, reps=8):
    """Measured device-copy bandwidth (read + write bytes per second of a large d2d copy): the achievable HBM figure on this box."""
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    del a, b
    return 2.0 * nbytes * reps / (ms * 1e-3) / 1e9


def plan_host(world, contexts, cores, quota, host_threads_arg=0):
    """Host threads per context and the way the library waits for the GPU, for `world` ranks of `contexts` contexts each on a host with
    `cores` CPUs of which the cgroup grants `quota` (None: all).  Pure function: tests/test_host_logic.py simulates the 8-rank launch.
    * threads: ranks x contexts x threads never exceeds what the process group may use (at least one thread per context, at most 8)
    * waits: hipEventSynchronize keeps ~5.5 CPUs per rank spinning (four contexts on cfg2); where the ranks together would spin on more
      than the quota the waits poll and nap (TRGT_POLL_WAIT: 2.2 CPUs per rank at 98 % of the rate with 200 us of spinning, 1.95 with none)"""
    cores_eff = max(1, min(cores, int(quota))) if quota else cores
    threads = min(8, host_threads_arg or max(1, cores_eff // max(1, world * contexts)))
    waits = {}
    if quota and world * 5.5 > quota:
        light = world * 2.3 <= quota
        waits = {"TRGT_POLL_WAIT": "1", "TRGT_POLL_SPIN_US": "200" if light else "0", "TRGT_POLL_NAP_US": "20" if light else "100"}
    return threads, waits


def cpu_quota():
    """CPUs this process may use at once: the cgroup's CFS quota (cpu.max: "quota period" or "max") when there is one, else None.  On the
    MI355X boxes of this pool os.cpu_count() is 256 and the quota 16: every host-side rate of this file (CPU baseline on "all cores",
    ingestion, writer) is a rate under that quota, and thread counts beyond ~2x the quota only add scheduling."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(path).read().split()
            if path.endswith("cpu.max"):
                if t[0] != "max":
                    return round(int(t[0]) / int(t[1]), 2)
            else:
                q = int(t[0])
                if q > 0:
                    return round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()), 2)
        except (OSError, ValueError, IndexError):
            pass
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="BASELINE workload; 0 (default) = config 2 (the one the metric is quoted on) followed, at N = 1, by short legs of configs 3, 4 and 5 reported under \"configs\" in the same JSON line")
    ap.add_argument("--loci", type=int, default=0, help="loci per GPU per step (default: by config)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streaming", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="with --config 0: config 2 only")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (synthetic BAM -> native ingestion -> GPU -> VCF / spanning BAM) reported under \"e2e\"")
    ap.add_argument("--e2e-loci", type=int, default=4000)
    ap.add_argument("--e2e-read-len", type=int, default=6000)
    ap.add_argument("--leg-steps", type=int, default=20, help="timed steps of each of the config 3 / 4 / 5 legs")
    ap.add_argument("--ws-limit-gb", type=float, default=0.0, help="workspace limit per context in GB (trgt_hip_set_workspace_limit; 0 = by config: the library's 32 GB, 8 GB for config 3)")
    ap.add_argument("--detail", default="", help="where the full record goes (default: bench_detail.json next to bench.py, and gpurun_out/ when it exists); stdout carries the compact contract line only")
    ap.add_argument("--contexts", type=int, default=0, help="contexts per GPU (trgt_hip_pool / trgt_locus_batch_many): worker threads, one context each, draining the queue of steps, so that the tail of one step (results back, host-path loci, HMM) overlaps the flank location of the next ones; 0 = by config (4; 6 for configs 3 and 5); 1 = the blocking call only")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the TRGT hot path")
    # (TRGT_BENCH_ONE_GPU=1: every rank on GPU 0 with the gloo backend -- the N > 1 code path on a one-GPU box, for testing only)
    one_gpu_test = os.environ.get("TRGT_BENCH_ONE_GPU") == "1"
    if one_gpu_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if one_gpu_test:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    # Waiting for the GPU: the library's default (hipEventSynchronize) keeps a CPU per context spinning -- 5.2 CPUs per rank with four
    # contexts on cfg2.  Where the cgroup grants fewer CPUs than the ranks of this job would spin on (the GPU boxes of this pool: 16),
    # the waits poll and nap instead (2.2 CPUs per rank at 98 % of the rate; 1.95 without any spinning): set before the library reads it.
    if "TRGT_POLL_WAIT" not in os.environ:
        for k, v in plan_host(world, max(1, args.contexts), os.cpu_count() or 8, cpu_quota())[1].items():
            os.environ.setdefault(k, v)
    env = dict(torch=torch, dist=dist, rank=rank, local_rank=local_rank, world=world)
    legs = args.config == 0 and not args.no_legs and world == 1
    first = argparse.Namespace(**vars(args))
    first.config = args.config or 2
    res = run_one(first, env)
    if legs:
        # BASELINE configs[2..4] under the same clock: short legs (HBM-resident value through the pool + the one-context loop the
        # roofline block is measured in + the parity sample), each reported in full under "configs"
        res["configs"] = {}
        for cfg in (3, 4, 5):
            a = argparse.Namespace(**vars(args))
            a.config, a.steps, a.warmup, a.no_streaming, a.loci, a.contexts, a.ws_limit_gb = cfg, args.leg_steps, min(args.warmup, 2), True, 0, 0, 0.0
            t0 = time.perf_counter()
            r = run_one(a, env, cpu_seconds=6.0, all_cores=False)
            r["leg_wall_s"] = round(time.perf_counter() - t0, 1)
            res["configs"][str(cfg)] = r
    if world == 1 and not args.no_e2e and args.config == 0:
        res["e2e"] = run_e2e(args, env)
    if rank == 0:
        # The full record (per-config detail, prose, every side figure) goes to a side file; stdout carries ONE compact contract line
        # (the single line had grown past what the driver keeps, and the round went unparsed).
        detail_path = write_detail(res, args.detail)
        line = compact_line(res, detail_path)
        print(line)
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


LINE_LIMIT = 4096   # bytes of the one stdout line (tests/test_bench_line.py)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "frac_io_only", "frac_traffic", "frac_at_reference_work", "bound_measured", "valu_issue_frac",
                 "avg_launch_ms", "launches", "algorithmic_bytes_per_launch", "dp_cells_per_launch")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")
CONFIG_KEYS = ("workload", "baseline_config", "value_streaming", "value_streaming_bam4", "value_single_context", "ms_per_step_single_context",
               "contexts_per_gpu", "loci_per_gpu", "reads_per_locus", "parallelism", "host_cpu_quota")
E2E_KEYS = ("ingest_loci_per_s", "ingest_loci_per_s_host", "gpu_loci_per_s", "gpu_loci_per_s_two_contexts", "gpu_loci_per_s_four_contexts", "write_loci_per_s", "write_loci_per_s_device_deflate",
            "pipeline_loci_per_s", "pipeline_loci_per_s_synchronous_writer", "pipeline_loci_per_s_device_ingest_host_deflate", "pipeline_loci_per_s_host_ingest", "pipeline_vcf_identical")


def _short(s, n):
    return s if len(s) <= n else s[:n - 1].rstrip() + "~"


def compact_line(res, detail_path=None):
    """The contract line: the keys the driver reads plus `roofline`, `cpu_baseline`, `parity`, one row per extra config and the
    end-to-end rates -- numbers and short labels only, guaranteed below LINE_LIMIT bytes (fields are dropped from the least important
    end if a future addition overshoots).  Pure function of the full record: tests/test_bench_line.py drives it without a GPU."""
    out = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "data"))
    out["vs_baseline"] = res.get("vs_baseline")
    out["dtype"] = "u8/u16 (WFA) + f64 (HMM)"
    cfg = _pick(res.get("config", {}), CONFIG_KEYS)
    if "workload" in cfg:
        cfg["workload"] = _short(cfg["workload"], 200)
    out["config"] = cfg
    roof = _pick(res.get("roofline", {}), ROOFLINE_KEYS)
    out["roofline"] = roof
    if res.get("cpu_baseline"):
        cb = _pick(res["cpu_baseline"], CPU_KEYS)
        cb["sample"] = _short(cb.get("sample", ""), 120)
        out["cpu_baseline"] = cb
    if res.get("cpu_baseline_all_cores"):
        out["cpu_baseline_all_cores"] = _pick(res["cpu_baseline_all_cores"], ("value", "cores", "cpu_quota"))
    if res.get("parity"):
        out["parity"] = _pick(res["parity"], ("parity_checked_loci", "mismatches"))
    if res.get("multi_gpu_digest_check"):
        out["multi_gpu_digest_check"] = res["multi_gpu_digest_check"]
    legs = {}
    for k, r in (res.get("configs") or {}).items():
        if not r:
            continue
        leg = _pick(r, ("value", "ms_per_step"))
        leg.update(_pick(r.get("config", {}), ("value_single_context", "ms_per_step_single_context", "loci_per_gpu")))
        leg["roofline"] = _pick(r.get("roofline", {}), ("kernel", "frac", "frac_io_only", "frac_traffic", "valu_issue_frac", "avg_launch_ms", "algorithmic_bytes_per_launch", "traffic", "bound_measured"))
        leg["cpu_baseline"] = (r.get("cpu_baseline") or {}).get("value")
        leg["parity"] = _pick(r.get("parity", {}), ("parity_checked_loci", "mismatches"))
        legs[k] = leg
    if legs:
        out["configs"] = legs
    if res.get("e2e"):
        out["e2e"] = _pick(res["e2e"], E2E_KEYS)
    if detail_path:
        out["detail"] = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path
    line = json.dumps(out, separators=(",", ":"))
    for drop in ("detail", "e2e", "cpu_baseline_all_cores", "configs"):  # never reached today; the contract keys are never dropped
        if len(line) < LINE_LIMIT:
            break
        out.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) < LINE_LIMIT, len(line)
    return line


def write_detail(res, path=None):
    """The whole record, pretty-printed, next to bench.py (bench_detail.json; also under gpurun_out/ when that exists, so that a
    gpurun call brings it back).  Returns the path written, or None when the directory is read-only."""
    path = path or os.path.join(ROOT, "bench_detail.json")
    written = None
    for p in (path, os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            if os.path.isdir(os.path.dirname(p)):
                with open(p, "w") as f:
                    json.dump(res, f, indent=1)
                    f.write("\n")
                written = written or p
        except OSError:
            pass
    return written


def run_e2e(args, env):
    """BAM -> VCF on this box: a synthetic coordinate-sorted BAM of full-length reads over cfg2-like loci (trgt_amd/synth_bam.py) through
    the native ingestion (trgt_ingest_batch_from_catalog), the locus path and the native VCF + spanning-BAM writer (trgt_writer_write).
    Ingestion two ways: the host path (.bai lookup, BGZF inflate with zlib, record decoding, clip_to_region on all host cores) and the
    device path (trgt_ingest_params.ingest_device, round 6: the compressed blocks go to the GPU, inflate + CRC-32 + record walk + clipping are
    kernels, only the clipped reads come back and the ASCII reads stay in HBM for the locus stage; INGEST_CALLERS host threads keep
    that many chunks in flight).  Every stage is timed on its own over the same chunks, then the three run as a pipeline (chunk queues
    between them): that wall time is what a user of the whole tool would see.  None of this is `value`."""
    import queue
    import shutil
    import tempfile
    import threading
    from trgt_amd import _lib, ingest, locus, shard, synth_bam, writers
    cores = os.cpu_count() or 8
    n, chunk = args.e2e_loci, 1000
    INGEST_CALLERS = int(os.environ.get("BENCH_INGEST_CALLERS", "3"))  # (a reader has six slots of device state: file read, upload, kernels and download of consecutive chunks overlap)
    INFLATE_WAVES = int(os.environ.get("BENCH_INFLATE_WAVES", "0"))  # (0: the library's default)
    WRITER_THREADS = int(os.environ.get("BENCH_WRITER_THREADS", "0"))  # (0: the library's default)
    WRITE_BEHIND = int(os.environ.get("BENCH_WRITE_BEHIND", "1"))
    d = tempfile.mkdtemp(prefix="trgt_e2e_")
    try:
        t0 = time.perf_counter()
        ds = synth_bam.write_dataset(d, n_loci=n, read_len=args.e2e_read_len)
        t_gen = time.perf_counter() - t0
        rd = ingest.Reader(ds["bam"], ds["fasta"])
        firsts = firsts_all = list(range(0, n, chunk))
        PIPE_PASSES = 3  # the pipelines walk the catalog this many times: with four chunks the fill and the drain of the three stages are half of the run
        dev = env["local_rank"]
        ing_threads = min(32, cores)  # host path, measured (tools/ingest_scaling.py): 1.3 k loci/s with 1 thread, 8.4 k with 8, 16 k with 16, 20 k with 32, 14 k with 64 -- under the box's CFS quota of 16 CPUs (cpu_quota())
        ing_host = lambda a, th=ing_threads: rd.batch(ds["bed"], first_locus=a, max_loci=chunk, keep_native=True, copy=False, read_names=False, threads=th, keep_bam4=1)
        ing_dev = lambda a: rd.batch(ds["bed"], first_locus=a, max_loci=chunk, keep_native=True, copy=False, read_names=False, threads=8, ingest_device=dev, inflate_waves_per_cu=INFLATE_WAVES)

        def ordered(fn, callers, firsts=None):
            """fn(first) over the chunks by `callers` threads, results in chunk order (a generator: chunk i is handed out as soon as it and all
            chunks before it are there; at most callers + 2 results wait)"""
            firsts = firsts if firsts is not None else firsts_all
            done, cv, st_, err = {}, threading.Condition(), dict(nxt=0, handed=0), []

            def work():
                while True:
                    with cv:
                        while st_["nxt"] < len(firsts) and st_["nxt"] - st_["handed"] > callers + 1 and not err:
                            cv.wait(0.01)
                        i = st_["nxt"]
                        st_["nxt"] += 1
                    if i >= len(firsts) or err:
                        return
                    try:
                        r = fn(firsts[i])
                    except Exception as e:  # noqa: BLE001
                        err.append(e)
                        r = None
                    with cv:
                        done[i] = r
                        cv.notify_all()
            th = [threading.Thread(target=work, daemon=True) for _ in range(callers)]
            for t in th:
                t.start()
            for i in range(len(firsts)):
                with cv:
                    while i not in done and not err:
                        cv.wait(0.05)
                    if err:
                        raise err[0]
                    r = done.pop(i)
                    st_["handed"] = i + 1
                    cv.notify_all()
                yield r
            for t in th:
                t.join()

        ing_host(0)  # page cache, thread start-up
        t0 = time.perf_counter()
        one = ing_host(0, 1)
        t_ing1 = (time.perf_counter() - t0) / max(1, one["n_loci"])
        t0 = time.perf_counter()
        batches = [ing_host(a) for a in firsts]
        t_ing_host = time.perf_counter() - t0
        # (the rates below are over PIPE_PASSES walks of the catalog: twelve chunks, not four -- with callers in flight the first chunk's
        #  latency is a quarter of a four-chunk run)
        t0 = time.perf_counter()
        for a in firsts * (PIPE_PASSES - 1):
            ing_host(a)
        t_ing_host = (t_ing_host + time.perf_counter() - t0) / PIPE_PASSES
        # ... the device path: one caller, then INGEST_CALLERS callers
        ing_dev(0)
        t0 = time.perf_counter()
        batches_d = [ing_dev(a) for a in firsts]
        t_ing_dev1 = time.perf_counter() - t0
        same_batches = all(int(x["n_reads"]) == int(y["n_reads"]) and np.array_equal(x["read_blob"], y["read_blob"]) and np.array_equal(x["read_off"], y["read_off"]) and
                           np.array_equal(x["qual_blob"], y["qual_blob"]) and np.array_equal(x["cigar"], y["cigar"]) and np.array_equal(x["mismatch_offsets"], y["mismatch_offsets"])
                           for x, y in zip(batches, batches_d))
        del batches_d
        t0 = time.perf_counter()
        for b in ordered(ing_dev, INGEST_CALLERS, firsts * PIPE_PASSES):
            del b
        t_ing_dev = (time.perf_counter() - t0) / PIPE_PASSES
        batches_d = list(ordered(ing_dev, INGEST_CALLERS))
        st = rd.device_stats()
        if st["fallbacks"]:
            raise SystemExit("bench.py: the device ingestion fell back to the host path (%r)" % (st,))
        views = [ingest.bam4_view(b) for b in batches]
        ctx = _lib.Context(env["local_rank"])
        params = locus.Params(host_threads=min(8, cores))
        gpu = lambda v: locus.run_batch(v, params, ctx)
        gpu_dev = lambda b: locus.run_batch(b, params, ctx, reads_dev=ingest.device_reads(b))  # the reads the ingestion left in HBM: nothing is uploaded
        for v in views[:2]:
            gpu(v)
        # (stage rates are the MEDIAN of PIPE_PASSES walks over the four chunks: one walk is 8 ms of GPU work, and a single scheduling hiccup
        #  of the box -- 10 to 30 ms, seen once in a dozen runs -- would be the whole figure; every walk's time goes to the detail record)
        def median_walk(fn):
            ts, last = [], None
            for _ in range(PIPE_PASSES):
                t0 = time.perf_counter()
                last = fn()
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2], last, [round(t, 5) for t in ts]
        t_gpu, outs, walks_gpu = median_walk(lambda: [gpu(v) for v in views])
        gpu_dev(batches_d[0])
        t_gpu_dev, outs_d, walks_gpu_dev = median_walk(lambda: [gpu_dev(b) for b in batches_d])
        for a, b2 in zip(outs, outs_d):
            if shard.result_digest(a, int(a.n_alleles.shape[0])) != shard.result_digest(b2, int(b2.n_alleles.shape[0])):
                raise SystemExit("bench.py: the e2e chunks from HBM-resident reads gave different results")
        del outs_d
        # ... and the same chunks through a pool of contexts draining them (trgt_locus_batch_many, what `value` is measured with)
        t_gpu_pool = {}
        for n_ctx in (2, 4):
            pl = _lib.Pool([env["local_rank"]] * n_ctx)
            try:
                locus.run_many(pl, views[:n_ctx], params)
                t_gpu_pool[n_ctx], (outs_p, _), _w = median_walk(lambda: locus.run_many(pl, views, params))
                for a, b2 in zip(outs, outs_p):
                    if shard.result_digest(a, int(a.n_alleles.shape[0])) != shard.result_digest(b2, int(b2.n_alleles.shape[0])):
                        raise SystemExit("bench.py: the e2e chunks through a pool gave different results")
                del outs_p
            finally:
                pl.close()
        def write_all(vcf, bam, bs, **kw):
            w = writers.Writer(rd, os.path.join(d, vcf), os.path.join(d, bam), threads=WRITER_THREADS, **kw)
            for b, o in zip(bs, outs):
                w.write(b, o)
            w.close()
        t_wr, _, walks_wr = median_walk(lambda: write_all("out.vcf", "out.spanning.bam", batches))
        # ... and with the BGZF blocks of the spanning BAM deflated on the GPU (trgt_writer_params.deflate_device, deflate_dev.hip; the
        # batches of the device path: the writer reads their pinned arrays)
        t_wr_dev, _, walks_wr_dev = median_walk(lambda: write_all("outd.vcf", "outd.spanning.bam", batches_d, deflate_device=env["local_rank"]))
        bam_bytes_host, bam_bytes_dev = os.path.getsize(os.path.join(d, "out.spanning.bam")), os.path.getsize(os.path.join(d, "outd.spanning.bam"))
        vcf_records = sum(1 for line in open(os.path.join(d, "out.vcf")) if not line.startswith("#"))
        # the genotypes against what the data set was made from (checked loosely here -- the parity proper is tests/ -- so that a broken
        # hand-over cannot report a rate)
        called, at = 0, 0
        for b, o in zip(views, outs):
            k = int(b["n_loci"])
            got = np.sort(o.allele_len[:2 * k].reshape(k, 2).astype(np.int64), axis=1)
            called += int(((got == np.sort(ds["allele_len"][at:at + k], axis=1)).all(axis=1) & (o.n_alleles[:k] == 2)).sum())
            at += k
        del outs, views, batches, batches_d

        # ---- pipeline: ingest | GPU | write
        stage_ms = {}

        def pipeline(tag, device_ingest, level, dev_deflate=-1, write_behind=0):
            w = writers.Writer(rd, os.path.join(d, tag + ".vcf"), os.path.join(d, tag + ".spanning.bam"), bam_compress_level=level, deflate_device=dev_deflate, threads=WRITER_THREADS,
                               write_behind=write_behind)
            q1, q2, err = queue.Queue(3), queue.Queue(2), []
            busy = dict(gpu=0.0, gpu_wait=0.0, write=0.0, write_wait=0.0)  # seconds a stage worked / waited for its input (the ingest stage is the callers')

            def stage_ingest():
                try:
                    for b in (ordered(ing_dev, INGEST_CALLERS, firsts * PIPE_PASSES) if device_ingest else (ing_host(a) for a in firsts * PIPE_PASSES)):
                        q1.put(b)
                except BaseException as e:  # noqa: BLE001
                    err.append(e)
                q1.put(None)

            def stage_gpu():
                try:
                    while True:
                        ta = time.perf_counter()
                        b = q1.get()
                        tb = time.perf_counter()
                        if b is None:
                            break
                        o = gpu_dev(b) if device_ingest else gpu(ingest.bam4_view(b))
                        busy["gpu"] += time.perf_counter() - tb
                        busy["gpu_wait"] += tb - ta
                        q2.put((b, o))
                except BaseException as e:  # noqa: BLE001
                    err.append(e)
                q2.put(None)
            t0 = time.perf_counter()
            th = [threading.Thread(target=stage_ingest, daemon=True), threading.Thread(target=stage_gpu, daemon=True)]
            for t in th:
                t.start()
            while True:
                ta = time.perf_counter()
                item = q2.get()
                tb = time.perf_counter()
                if item is None:
                    break
                w.write(*item)
                busy["write"] += time.perf_counter() - tb
                busy["write_wait"] += tb - ta
            w.close()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            if err:
                raise err[0]
            body = lambda path: [line for line in open(path) if not line.startswith("#")]
            stage_ms[tag] = {k: round(1e3 * v / (len(firsts) * PIPE_PASSES), 2) for k, v in busy.items()}  # per chunk
            return dt / PIPE_PASSES, body(os.path.join(d, "out.vcf")) * PIPE_PASSES == body(os.path.join(d, tag + ".vcf"))

        t_pipe_host, same = pipeline("out2", False, 6)            # as rounds 1-5 measured it: host ingestion, htslib's BAM level, zlib
        t_pipe_hostd, same_b = pipeline("out3", False, 6, dev)    # ... the spanning BAM deflated on the GPU
        t_pipe_dev6, same_c = pipeline("out4", True, 6)           # device ingestion, the spanning BAM by zlib at level 6
        t_pipe_sync, same_e = pipeline("out6", True, 6, dev)      # device ingestion + device deflate: the host only reads the file, formats records and writes
        # ... and with the writer's write-behind (ABI 11): a batch is deflated and written while the next one is formatted -- the pipeline's figure
        t_pipe, same_d = pipeline("out5", True, 6, dev, WRITE_BEHIND)
        same = same and same_e
        same = same and same_b and same_c and same_d
        r = lambda x: round(x, 1)
        return dict(
            workload="%d cfg2-like loci (motif 2-6 bp, 5-40 copies per allele), %d reads of ~%d bases per locus, one contig; BAM %.1f MB (%d reads, %.0f MB of records), written by trgt_amd/synth_bam.py in %.1f s"
                     % (n, 30, args.e2e_read_len, ds["bam_bytes"] / 1e6, ds["n_reads"], ds["bases"] / 1e6, t_gen),
            chunk_loci=chunk, pipeline_passes=PIPE_PASSES, ingest_threads=ing_threads, ingest_callers=INGEST_CALLERS, host_cores=cores, host_cpu_quota=cpu_quota(),
            ingest_loci_per_s=r(n / t_ing_dev), ingest_loci_per_s_one_caller=r(n / t_ing_dev1), ingest_loci_per_s_host=r(n / t_ing_host), ingest_loci_per_s_host_one_thread=r(1.0 / t_ing1),
            ingest_record_mb_per_s=r(ds["bases"] / 1e6 / t_ing_dev), ingest_device_same_batches=bool(same_batches), ingest_device_stats=st,
            gpu_loci_per_s=r(n / t_gpu_dev), gpu_loci_per_s_host_reads=r(n / t_gpu), gpu_loci_per_s_two_contexts=r(n / t_gpu_pool[2]), gpu_loci_per_s_four_contexts=r(n / t_gpu_pool[4]),
            write_loci_per_s=r(n / t_wr), write_loci_per_s_device_deflate=r(n / t_wr_dev),
            stage_walk_s=dict(gpu=walks_gpu_dev, gpu_host_reads=walks_gpu, write=walks_wr, write_device_deflate=walks_wr_dev),  # (every walk over the four chunks; the rates above are the medians)
            pipeline_loci_per_s=r(n / t_pipe), pipeline_s=round(t_pipe, 3), pipeline_write_behind=bool(WRITE_BEHIND), pipeline_loci_per_s_synchronous_writer=r(n / t_pipe_sync),
            pipeline_loci_per_s_device_ingest_host_deflate=r(n / t_pipe_dev6),
            pipeline_loci_per_s_host_ingest=r(n / t_pipe_host), pipeline_loci_per_s_host_ingest_device_deflate=r(n / t_pipe_hostd),
            spanning_bam_mb=round(bam_bytes_host / 1e6, 1), spanning_bam_mb_device_deflate=round(bam_bytes_dev / 1e6, 1),
            pipeline_stage_ms_per_chunk=dict(host_ingest=stage_ms["out2"], host_ingest_device_deflate=stage_ms["out3"], device_ingest_host_deflate=stage_ms["out4"], device_ingest_device_deflate=stage_ms["out6"],
                                             device_ingest_device_deflate_write_behind=stage_ms["out5"]),
            vcf_records=vcf_records, loci_with_both_true_allele_lengths=called, pipeline_vcf_identical=bool(same),
            bound="the three stages are within 2 ms per chunk of each other with the writer's write-behind (synchronous, the writer's record formatting on the host cores of the quota is the busiest); the ingestion: the device inflate is one wave per BGZF block and a launch lasts ceil(blocks / resident waves) block times of ~ 3.8 ms; see DESIGN.md")
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run_one(args, env, cpu_seconds=20.0, all_cores=True):
    """One BASELINE config measured on this rank's GPU; returns the result dict on rank 0 (None elsewhere)."""
    torch, dist, rank, local_rank, world = env["torch"], env["dist"], env["rank"], env["local_rank"], env["world"]
    n_loci = args.loci or DEFAULT_LOCI[args.config]
    if args.contexts <= 0:
        # measured on MI355X (tools/contexts_sweep.sh): config 2 is flat from 4 contexts on (1.87 M loci/s with 4, 5, 6 and 8); config 5,
        # whose calls wait on the host between their GPU stages, still gains (210 / 235 / 245 k with 4 / 6 / 8; 10 do not fit the HBM any more);
        # config 4 is limited by its workspaces (tens of GB per context), config 3 runs six contexts with a smaller workspace limit (below).
        # value_single_context is printed next to value
        args.contexts = {3: 6, 5: 6}.get(args.config, 4)
    if args.ws_limit_gb <= 0 and args.config == 3:
        # the generic alignment kernel sizes its HBM arena by this limit; since the long reads meet the pre-filter window by window it has
        # little left to do, and six contexts of 12 GB do more than three of 32 (8.2 against 6.4 k loci/s; 16 GB the same, 8 GB 7.6 k)
        args.ws_limit_gb = 12.0
    ws_limit = int(args.ws_limit_gb * (1 << 30))

    from trgt_amd import _lib, locus, shard

    # ---- synthetic shard of this rank (untimed)
    # host threads for the glue between the GPU stages; the library uses at most 8 of them when the reads are resident in HBM
    # (ranks x contexts per GPU x host threads per context never exceeds the host's cores: the first real 8-GPU run must not oversubscribe)
    cores = os.cpu_count() or 8
    quota = cpu_quota()
    host_threads = plan_host(world, args.contexts, cores, quota, args.host_threads)[0]  # (the cgroup lets this process use 16 of the 256 CPUs of the GPU boxes)
    batch = make_batch(args.config, n_loci, rank * n_loci)
    reads_dev = torch.from_numpy(batch["read_blob"]).cuda()
    flank_dev = torch.from_numpy(batch["flank_blob"]).cuda()
    ctx = _lib.Context(local_rank)
    params = locus.Params(host_threads=host_threads)
    out = locus.BatchOutputs(batch)

    def step():
        locus.run_batch(batch, params, ctx, out, flank_dev=flank_dev, reads_dev=reads_dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()  # setup: primes the library's device-buffer pool and code objects (part of the untimed set-up, like the H2D upload)
    ctx.timing_enable(True)  # per-kernel HIP events (roofline).  The HIP runtime spends a one-off ~9 ms in the second call after the
    step()                   # first timing event is recorded on a stream (measured: tools/step_times.py timing), so two more
    step()                   # set-up calls run with the events on before the warm-up proper
    for _ in range(args.warmup):
        step()
    ctx.timing_reset()
    import gc
    gc.collect()
    gc.disable()  # a generation-2 collection of the driver script's own objects showed up as a 30-40 ms pause in one call out of ~400
    # ---- K steps on ONE context (the blocking call, batch after batch) ...
    fence()
    t0 = time.perf_counter()
    marks = [t0]
    for _ in range(args.steps):
        step()
        marks.append(time.perf_counter())  # (a call is synchronous: no extra synchronisation is added inside the timed region)
    fence()
    dt_single = time.perf_counter() - t0
    names = {k: n for n, k in (("flank_scan", 0), ("wfa_consensus", 1), ("hmm_viterbi", 2), ("wfa_flank", 3), ("wfa_flank_rest", 4), ("wfa_filter", 5), ("flank_window", 6))}
    kt = {names[k]: ctx.timing_get(k) for k in names}
    ctx.timing_enable(False)
    dt_single_instrumented = dt_single
    # ... and the same loop once more WITHOUT the timing events: recorded on the side streams of the HMM launches they keep launches
    # that would run next to each other from overlapping (config 3: 43 against 30 ms per call), so the one-context figure that is
    # reported is this one; the loop above is only where the per-kernel times (roofline) come from
    n_plain = max(4, min(args.steps, 50))
    step(); step()
    fence()
    t0 = time.perf_counter()
    for _ in range(n_plain):
        step()
    fence()
    dt_single = (time.perf_counter() - t0) * args.steps / n_plain
    dt = dt_single
    # ---- ... and through `contexts` worker threads, one context each, all on this rank's GPU, draining a queue of the same K steps:
    #      a call's host-bound tail (results back, the few loci of the host path, HMM collection) and the kernels of its last stage
    #      overlap the flank location of the next calls.  This is `value`; the per-kernel times below are then those of this region,
    #      summed over the contexts.
    # ---- ... and through `contexts` contexts on this rank's GPU behind one queue (trgt_hip_pool / trgt_locus_batch_many: one worker
    #      thread per context inside the library), draining the same K steps: a call's tail (results back, the few loci of the host
    #      path, HMM collection) and the kernels of its last stage overlap the flank location of the next calls.  This is `value`; the
    #      per-kernel times below are then those of this region, summed over the contexts.
    pool, kt_pool, cpus_busy = None, None, None
    if args.contexts > 1:
        pool = _lib.Pool([local_rank] * args.contexts)
        if ws_limit:
            for pc in pool.contexts:
                pc.check(_lib.lib().trgt_hip_set_workspace_limit(pc.handle, ws_limit))
        outs_w = [locus.BatchOutputs(batch) for _ in range(args.contexts)]
        many = lambda n: locus.run_many(pool, [batch] * n, params, outs_w, flank_dev=flank_dev, reads_dev=reads_dev, out_per_context=True)
        # set-up of every context (buffer pools, code objects, first touch of its workspaces): at least 3 batches per context and 0.6 s
        # (a fresh context runs at half speed for its first few hundred milliseconds on the host-heavy configs)
        t_w = time.perf_counter()
        many(3 * args.contexts)
        while time.perf_counter() - t_w < 0.6:
            many(2 * args.contexts)
        fence()
        import resource
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        _, ran = many(args.steps)  # (no timing events in this region: pure throughput)
        fence()
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        cpus_busy = (ru1.ru_utime - ru0.ru_utime + ru1.ru_stime - ru0.ru_stime) / max(dt, 1e-9)  # host CPUs this rank kept busy while `value` was measured
        # per-kernel times of the same region, from a shorter instrumented pass
        for c in pool.contexts:
            c.timing_enable(True)
        many(3 * args.contexts)  # (the one-off cost of the first timing events, see above)
        for c in pool.contexts:
            c.timing_reset()
        n_instr = max(args.contexts, min(args.steps, 24))
        many(n_instr)
        kt_pool = {names[k]: tuple(sum(c.timing_get(k)[i] for c in pool.contexts) * (args.steps / n_instr if i == 0 else 1) for i in range(3)) for k in names}
        for w in set(ran):
            if shard.result_digest(outs_w[w], n_loci) != shard.result_digest(out, n_loci):
                raise SystemExit("bench.py: a pool context returned different results")
        for c in pool.contexts:
            c.timing_enable(False)
    gc.enable()
    step_ms = sorted(1e3 * (b - a) for a, b in zip(marks, marks[1:]))
    dt = shard.max_over_ranks(dt, dist if world > 1 else None, device="cuda")
    dt_single = shard.max_over_ranks(dt_single, dist if world > 1 else None, device="cuda")

    # ---- the same batches with the reads starting in pinned host memory, through the pipelined entry points: the upload of batch
    #      k + 1 (trgt_locus_batch_submit, copy stream) runs next to the kernels of batch k (trgt_locus_batch_wait)
    dt_stream = dt_stream_single = dt_stream4 = dt_stream4_single = None
    if not args.no_streaming:
        pins = [torch.from_numpy(batch["read_blob"]).pin_memory() for _ in range(2)]
        outs_s = [locus.BatchOutputs(batch) for _ in range(2)]
        n_s = max(4, min(args.steps, 40))

        def stream_loop(n):
            t = locus.submit_batch(batch, params, ctx, outs_s[0], flank=flank_dev, reads=pins[0])
            for k in range(n):
                nxt = locus.submit_batch(batch, params, ctx, outs_s[(k + 1) % 2], flank=flank_dev, reads=pins[(k + 1) % 2]) if k + 1 < n else None
                t.wait()
                t = nxt

        stream_loop(3)
        gc.collect()
        gc.disable()
        fence()
        t0 = time.perf_counter()
        stream_loop(n_s)
        fence()
        dt_stream = (time.perf_counter() - t0) / n_s
        gc.enable()
        dt_stream = shard.max_over_ranks(dt_stream, dist if world > 1 else None, device="cuda")
        for o in outs_s:
            if shard.result_digest(o, n_loci) != shard.result_digest(out, n_loci):
                raise SystemExit("bench.py: host-resident reads gave different results than HBM-resident reads")
        dt_stream_single = dt_stream
        if pool is not None:  # ... and through the pool, every context running the blocking call on reads in pinned host memory (its
            # own upload, then its kernels): the upload of one call runs next to the kernels of the others.  (One pinned copy: the
            # contexts only read it.)
            outs_sw = [locus.BatchOutputs(batch) for _ in range(args.contexts)]
            smany = lambda n: locus.run_many(pool, [batch] * n, params, outs_sw, flank_dev=flank_dev, reads_dev=pins[0], out_per_context=True)
            smany(3 * args.contexts)
            gc.collect()
            gc.disable()
            fence()
            t0 = time.perf_counter()
            _, sran = smany(n_s)
            fence()
            dt_stream = shard.max_over_ranks((time.perf_counter() - t0) / n_s, dist if world > 1 else None, device="cuda")
            gc.enable()
            for w in set(sran):
                if shard.result_digest(outs_sw[w], n_loci) != shard.result_digest(out, n_loci):
                    raise SystemExit("bench.py: host-resident reads gave different results than HBM-resident reads")

        # ... and with the reads as BAM 4-bit codes (TRGT_READS_BAM4: what a BAM record holds; half the bytes cross the link, one kernel
        #     expands them in HBM), pinned as well
        pk = locus.pack_bam4(batch, pinned=True)

        def stream_loop4(n):
            t = locus.submit_batch(pk, params, ctx, outs_s[0], flank=flank_dev)
            for k in range(n):
                nxt = locus.submit_batch(pk, params, ctx, outs_s[(k + 1) % 2], flank=flank_dev) if k + 1 < n else None
                t.wait()
                t = nxt

        stream_loop4(3)
        gc.collect()
        gc.disable()
        fence()
        t0 = time.perf_counter()
        stream_loop4(n_s)
        fence()
        dt_stream4 = dt_stream4_single = shard.max_over_ranks((time.perf_counter() - t0) / n_s, dist if world > 1 else None, device="cuda")
        gc.enable()
        for o in outs_s:
            if shard.result_digest(o, n_loci) != shard.result_digest(out, n_loci):
                raise SystemExit("bench.py: 4-bit reads gave different results than ASCII reads")
        if pool is not None:
            smany4 = lambda n: locus.run_many(pool, [pk] * n, params, outs_sw, flank_dev=flank_dev, out_per_context=True)
            smany4(3 * args.contexts)
            gc.collect()
            gc.disable()
            fence()
            t0 = time.perf_counter()
            _, sran = smany4(n_s)
            fence()
            dt_stream4 = shard.max_over_ranks((time.perf_counter() - t0) / n_s, dist if world > 1 else None, device="cuda")
            gc.enable()
            for w in set(sran):
                if shard.result_digest(outs_sw[w], n_loci) != shard.result_digest(out, n_loci):
                    raise SystemExit("bench.py: 4-bit reads gave different results than ASCII reads")

    # ---- N > 1: every rank recomputes its right neighbour's shard; the digests must agree (N-GPU output == 1-GPU output)
    digest_check = None
    if world > 1:
        mine = shard.result_digest(out, n_loci)
        nb = (rank + 1) % world
        nbatch = make_batch(args.config, n_loci, nb * n_loci)
        nout = locus.run_batch(nbatch, params, ctx, flank_dev=torch.from_numpy(nbatch["flank_blob"]).cuda(), reads_dev=torch.from_numpy(nbatch["read_blob"]).cuda())
        theirs = shard.result_digest(nout, n_loci)
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, theirs))
        bad = [r for r in range(world) if gathered[r][1] != gathered[(r + 1) % world][0]]
        if bad:
            raise SystemExit("bench.py: shard digests differ between GPUs (ranks %s recomputed their neighbour's shard differently)" % bad)
        digest_check = {"ranks": world, "shards_recomputed_on_another_gpu": world, "digest_mismatches": 0}

    if pool is not None:
        pool.close()
    res = None
    if rank == 0:
        # wfa_filter: register-resident pre-filter over the alignments of reads too short to span their locus (>90 % of the wavefront
        # offsets); wfa_flank: the back-tracing kernel over the alignments the filter keeps; wfa_flank_rest: the other flank alignments
        # (on seeded windows, then the few that need the whole read); flank_scan: exact-match scan + segment search for the windows
        # (flank_window is left out of the choice: the seed-search launches sit on the stream next to the pre-filter's and their events
        #  bracket the WAIT for its persistent workgroups -- 0.1 ms of work inside up to 2.5 ms, profiles/*kernel_stats_one_context.txt.
        #  Counted under flank_scan, as until ABI 9, that wait made the scan "dominant" in one run out of a few, by a hair.)
        dom = max((k for k in kt if k != "flank_window"), key=lambda k: kt[k][0])
        ms, launches, cells = kt[dom]
        stats = out.stats
        n_reads = int(batch["n_reads"])
        mean_read = float(batch["read_len"].mean())
        launches = max(launches, 1)
        cells_l = cells / launches
        # ALGORITHMIC bytes per launch of the dominant kernel (DESIGN.md "Roofline model", SURVEY.md 8(d): B_io + B_dp)
        if dom == "wfa_filter":     # B_dp = 4 B per wavefront offset; the kernel keeps that state in registers: B_io is all it moves
            jobs = int(stats[14])
            io_bytes = jobs * (250 + mean_read + 12)
            dp_bytes = 4.0 * cells_l
        elif dom == "wfa_flank":    # 2 B per offset of history actually written + sequences in + (n_match, span) out
            jobs = int(stats[16]) or int(stats[14]) or int(stats[0])
            io_bytes = jobs * (250 + mean_read + 20)
            dp_bytes = 4.0 * cells_l
        elif dom == "hmm_viterbi":  # 1 B per back-pointer cell + allele in + annotation out
            io_bytes = float(out.allele_len.sum()) * 2
            dp_bytes = 1.0 * cells_l
        elif dom == "flank_scan":   # every read byte once + 4 B per (read, side)
            io_bytes = float(batch["read_len"].sum()) + 8.0 * n_reads
            dp_bytes = 0.0
        else:
            io_bytes = 0.0
            dp_bytes = 4.0 * cells_l
        bytes_per_launch = io_bytes + dp_bytes
        traffic = None  # measured HBM bytes per launch of the same kernel / workload, when a PMC profile is committed
        valu_insts = None  # VALU wave-instructions per launch of the same kernel, from the committed SQ counters
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            tj = tj.get("configs", {}).get(str(args.config), tj)  # (config 2 at the top level, the others under "configs")
            if tj.get("loci_per_gpu") == n_loci and tj.get("config", args.config) == args.config and dom in tj["kernels"]:
                traffic = int(tj["kernels"][dom]["bytes_per_launch"])
                valu_insts = tj["kernels"][dom].get("valu_wave_insts_per_launch")
        except (OSError, ValueError, KeyError):
            pass
        avg_ms = ms / launches
        gbs = lambda b: b / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        achieved = gbs(bytes_per_launch)
        # The pre-filter gives an alignment up once it cannot reach the match threshold any more: `cells` are the offsets it computed.
        # The offsets WFA2-lib computes for the same alignments (every one run to its end) come from one untimed call with that off.
        ref_cells_l = None
        if dom == "wfa_filter":
            rctx = _lib.context_with_env(device=local_rank, TRGT_WFA_NO_EARLY=1)
            rout = locus.BatchOutputs(batch)
            locus.run_batch(batch, params, rctx, rout, flank_dev=flank_dev, reads_dev=reads_dev)
            ref_cells_l = int(rout.stats[17])
            rctx.close()
        copy_peak = copy_peak_gbs(torch)
        num_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        # what binds the dominant kernel: VALU issue.  A wave-instruction occupies its SIMD's 16 lanes for 4 cycles; the
        # pass has num_cus * 4 SIMDs for avg_ms: issue fraction = committed VALU wave-instructions of one launch * 4 / SIMD-cycles of it
        valu_issue_frac = round(valu_insts * 4.0 / (num_cus * 4 * avg_ms * 1e-3 * CLOCK_GHZ * 1e9), 4) if valu_insts and avg_ms > 0 else None
        valu_peak = num_cus * VALU_LANES_PER_CU * CLOCK_GHZ * 1e9   # 32-bit integer lane-operations per second
        cells_per_s = cells / max(ms, 1e-9) * 1e3
        res = {
            "metric": "loci/s", "value": round(world * n_loci * args.steps / dt, 1), "unit": "loci/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "ms_per_step_single_context_min_median_max": [round(step_ms[0], 2), round(step_ms[len(step_ms) // 2], 2), round(step_ms[-1], 2)],  # rank 0
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8+u8 packed (WFA pre-filter), u16 (WFA back-trace), f64 (HMM)",
            "data": "synthetic",
            "value_is": "HBM-resident, the K steps REPLAY ONE BATCH (the contexts of the pool read the same blob: the L2 / MALL see it more than once; nothing in the step is HBM-bound, and no result is kept between steps): read and flank bytes are in HBM before the timed region; value_streaming: every batch's reads start in pinned host memory and cross PCIe inside the timed region -- through the same worker contexts, each uploading its batch and then computing on it (single context: uploaded by trgt_locus_batch_submit next to the compute of the batch before, trgt_locus_batch_wait); value_streaming_bam4: the same with the reads as BAM 4-bit codes (TRGT_READS_BAM4), expanded in HBM",
            "value_streaming": round(world * n_loci / dt_stream, 1) if dt_stream else None,
            "value_streaming_single_context": round(world * n_loci / dt_stream_single, 1) if dt_stream else None,
            "value_streaming_bam4": round(world * n_loci / dt_stream4, 1) if dt_stream4 else None,
            "value_streaming_bam4_single_context": round(world * n_loci / dt_stream4_single, 1) if dt_stream4 else None,
            "value_single_context": round(world * n_loci * args.steps / dt_single, 1), "ms_per_step_single_context": round(1e3 * dt_single / args.steps, 3), "ms_per_step_single_context_with_timing_events": round(1e3 * dt_single_instrumented / args.steps, 3),
            "config": {"workload": WORKLOAD[args.config] % n_loci, "baseline_config": args.config,
                       # (the figures next to `value` that a reader of the parsed line needs: SURVEY 8(d)'s "first H2D to last D2H" rates and the blocking call)
                       "value_streaming": round(world * n_loci / dt_stream, 1) if dt_stream else None,
                       "value_streaming_bam4": round(world * n_loci / dt_stream4, 1) if dt_stream4 else None,
                       "value_single_context": round(world * n_loci * args.steps / dt_single, 1), "ms_per_step_single_context": round(1e3 * dt_single / args.steps, 3),
                       "steps_replay_one_batch": True,
                       "loci_per_gpu": n_loci, "reads_per_locus": 30, "parallelism": "loci sharded across %d GPU(s), no collective" % world,
                       "host_threads_per_context": host_threads, "contexts_per_gpu": args.contexts, "host_cores": cores, "host_cpu_quota": cpu_quota(),
                       "host_threads_all_ranks": host_threads * args.contexts * world, "gpu_waits": ("poll, spin %s us, nap %s us" % (os.environ.get("TRGT_POLL_SPIN_US", "2000"), os.environ.get("TRGT_POLL_NAP_US", "20"))) if os.environ.get("TRGT_POLL_WAIT", "0") not in ("", "0") else "hipEventSynchronize (spins)", "host_cpus_busy_per_rank": round(cpus_busy, 2) if cpus_busy is not None else None, "workspace_limit_gb_per_context": args.ws_limit_gb or 32.0},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         # what the HBM counters saw (profiles/traffic.json: FETCH_SIZE + WRITE_SIZE per launch) as a fraction of peak: the utilisation proper
                         "frac_traffic": round(gbs(traffic) / HBM_PEAK_GBS, 6) if traffic else None,
                         # the bound that binds, and the fractions that say the HBM pricing is nominal
                         "bound_measured": "valu-issue" if dom in ("wfa_filter", "wfa_flank", "wfa_flank_rest", "flank_scan") else ("scalar-issue / latency" if dom == "wfa_consensus" else "latency (dependent chain per column)"),
                         "valu_issue_frac": valu_issue_frac, "valu_wave_insts_per_launch": valu_insts,
                         "frac_at_reference_work": round(gbs(io_bytes + 4.0 * ref_cells_l) / HBM_PEAK_GBS, 5) if ref_cells_l else None,
                         "avg_launch_ms": round(avg_ms, 3), "launches": int(launches), "measured_in": "the single-context loop of this run (K steps, HIP events on the kernel's stream)",
                         "launch_is": "wfa_filter runs as two kernel launches over one job list when the longest text needs more than 1024 diagonals (wfa_filter_kernel<9,1> or <5,2> / <6,2> for the jobs above 1024, then <4,2>): a 'launch' here is the pair, one pass of the pre-filter over a batch.  Outside a pool the two kernels run NEXT TO each other on two streams (round 3): the pass lasts from the start of the first to the end of the last -- in a rocprofv3 kernel summary of the one-context command that is about the longer kernel's average, not the sum (TRGT_FILTER_SERIAL=1 puts them one after the other again: then the sum)" if dom == "wfa_filter" else None,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "algorithmic_bytes_model": "SURVEY.md 8(d): B_io + B_dp, B_dp = 4 B per wavefront offset W (1 B per Viterbi cell); W = offsets the kernel computed, counted on the device (the pre-filter stops an alignment that cannot reach the match threshold: W is below WFA2-lib's count, see dp_cells_reference_per_launch)",
                         "dp_cells_reference_per_launch": ref_cells_l,
                         "achieved_at_reference_work": round(gbs(io_bytes + 4.0 * ref_cells_l), 2) if ref_cells_l else None,
                         "frac_io_only": round(gbs(io_bytes) / HBM_PEAK_GBS, 7), "io_bytes_per_launch": int(io_bytes),
                         "peak_measured_copy": round(copy_peak, 1), "frac_of_measured_copy": round(achieved / copy_peak, 5) if copy_peak > 0 else None,
                         "note": "the pre-filter keeps the wavefront state (B_dp) in registers: frac prices it as if it were streamed, frac_io_only is what actually moves",
                         "dp_cells_per_launch": int(cells_l), "dp_cells_per_s": round(cells_per_s, 1)},
            # the informative bound for this integer DP: DP offsets per second against the VALU issue peak
            "issue_roofline": {"kernel": dom, "dp_offsets_per_s": round(cells_per_s, 1), "valu_lane_ops_peak_per_s": valu_peak,
                               "offsets_per_lane_op_at_peak": round(cells_per_s / valu_peak, 5),
                               "note": "peak = CUs x 64 lanes x 2.4 GHz; lane-operations per offset and VALU utilisation from the committed SQ counters (profiles/, DESIGN.md)"},
            # per-kernel HIP-event times: of the single-context loop (a kernel's own duration inside its call: what the roofline block
            # uses) and of the region `value` is measured on, summed over the contexts (there a kernel's events also bracket the time it
            # shares the CUs with the other contexts' kernels)
            "kernels_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in kt.items()},
            "kernels_ms_per_step_value_region": {k: round(v[0] / args.steps, 3) for k, v in kt_pool.items()} if kt_pool else None,
            # host-visible wall time of the last step: blocked on stage A + device genotyper; consensus alignments of the loci
            # handed back to the host path; HMM enqueue / collect (its kernel overlaps the host path); host glue; whole call
            "stage_ms_last_step": {"wait_flank_location_and_genotyper": round(stats[4] / 1e6, 2), "consensus": round(stats[5] / 1e6, 2),
                                   "hmm_host_visible": round(stats[6] / 1e6, 2), "host_glue": round(stats[7] / 1e6, 2),
                                   "total": round(stats[8] / 1e6, 2)},
            "work_per_step": {"flank_wfa_jobs": int(stats[0]), "flank_wfa_jobs_expensive": int(stats[14]), "filter_kept": int(stats[16]),
                              "filter_offsets": int(stats[17]), "consensus_jobs": int(stats[1]), "edit_distance_jobs": int(stats[15]),
                              "spanning_reads": int(stats[2]), "hmm_jobs": int(stats[3])},
        }
        if digest_check:
            res["multi_gpu_digest_check"] = digest_check
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N = 1 only
            res["cpu_baseline"], res["parity"] = cpu_baseline(batch, out, args.config, seconds_budget=cpu_seconds)
            nthr = min(os.cpu_count() or 1, 128)
            if all_cores and nthr > 1 and args.config in (2, 4):
                res["cpu_baseline_all_cores"] = cpu_baseline_mt(batch, nthr)
    ctx.close()
    del reads_dev, flank_dev
    torch.cuda.empty_cache()
    return res if rank == 0 else None


if __name__ == "__main__":
    main()
