#!/usr/bin/env python3
"""bench.py -- loci/s of the TRGT hot path on MI355X (BASELINE.json metric), one process per GPU.

A "step" = one full pass of trgt_locus_batch over one batch of synthetic loci (BASELINE.json configs[1]:
10k single-motif STR loci, motif 3-6 bp, allele <= 200 bp, 30x HiFi) per GPU:
flank location (exact scan + ends-free WFA fallback) -> host length genotyping -> consensus BiWFA where
needed -> motif-HMM labelling (MS/MC/AP).  Read and flank bytes are resident in HBM before the timed
region; only the per-read offsets / spans / alleles cross PCIe inside it.

  python bench.py [--gpus 1] [--steps 3] [--warmup 1] [--loci 10000]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Loci shard embarrassingly: rank r generates and processes loci [r*L, (r+1)*L) (weak scaling, no
collective on the data path); value = loci of all ranks * steps / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md; ~6.3 TB/s achievable)


def cpu_baseline(batch, seconds_budget=20.0, max_loci=8000):
    """The CPU oracle (port of the reference algorithms, single thread) timed on a bounded sample of the same batch."""
    from oracle import binding as orc
    orc.lib()
    n = min(int(batch["n_loci"]), max_loci)
    t0 = time.perf_counter()
    done = 0
    for l in range(n):
        a0, a1 = int(batch["locus_read_begin"][l]), int(batch["locus_read_begin"][l + 1])
        reads = [bytes(batch["read_blob"][int(batch["read_off"][r]):int(batch["read_off"][r]) + int(batch["read_len"][r])])
                 for r in range(a0, a1)]
        lf = bytes(batch["flank_blob"][int(batch["lf_off"][l]):int(batch["lf_off"][l]) + int(batch["lf_len"][l])])
        rf = bytes(batch["flank_blob"][int(batch["rf_off"][l]):int(batch["rf_off"][l]) + int(batch["rf_len"][l])])
        tr = bytes(batch["tr_blob"][int(batch["tr_off"][l]):int(batch["tr_off"][l]) + int(batch["tr_len"][l])])
        m0, m1 = int(batch["set_motif_begin"][l]), int(batch["set_motif_begin"][l + 1])
        motifs = [bytes(batch["motif_blob"][int(batch["motif_off"][m]):int(batch["motif_off"][m + 1])]) for m in range(m0, m1)]
        orc.locus_analyze(lf, rf, tr, motifs, reads)
        done += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    return dict(value=round(done / dt, 2), unit="loci/s", cores=1, kind="port",
                sample="first %d loci of the same synthetic batch, oracle/liboracle.so (C++ restatement), 1 thread, %.1f s" % (done, dt))


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
    return dict(value=round(done / dt, 2), unit="loci/s", cores=threads, kind="port",
                sample="all %d loci of the same synthetic batch, oracle/liboracle.so, %d native threads pulling chunks of loci from a shared counter, %.1f s" % (done, threads, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--loci", type=int, default=10000, help="loci per GPU per step")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the TRGT hot path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from trgt_amd import _lib, locus, shard, synth

    # ---- synthetic shard of this rank (untimed)
    # host threads for the glue between the GPU stages; the library uses at most 8 of them when the reads are resident in HBM
    host_threads = min(8, args.host_threads or max(1, (os.cpu_count() or 8) // max(1, world)))
    batch = synth.generate(args.loci, first_locus=rank * args.loci, config=2)
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
    fence()
    t0 = time.perf_counter()
    marks = [t0]
    for _ in range(args.steps):
        step()
        marks.append(time.perf_counter())  # (a call is synchronous: no extra synchronisation is added inside the timed region)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    step_ms = sorted(1e3 * (b - a) for a, b in zip(marks, marks[1:]))
    ctx.timing_enable(False)
    dt = shard.max_over_ranks(dt, dist if world > 1 else None, device="cuda")

    if rank == 0:
        # wfa_flank: the launch over the alignments of reads too short to span their locus (95 % of the wavefront offsets);
        # wfa_flank_rest: the launches over the other flank alignments (on seeded windows, then the few that need the whole read);
        # flank_scan: the exact-match scan and the segment search for the windows
        names = {_lib_k: n for n, _lib_k in (("flank_scan", 0), ("wfa_consensus", 1), ("hmm_viterbi", 2), ("wfa_flank", 3), ("wfa_flank_rest", 4), ("wfa_filter", 5))}
        kt = {names[k]: ctx.timing_get(k) for k in names}
        dom = max(kt, key=lambda k: kt[k][0])
        ms, launches, cells = kt[dom]
        stats = out.stats
        n_reads = int(batch["n_reads"])
        # ALGORITHMIC bytes per launch of the dominant kernel (DESIGN.md "Roofline model"):
        if dom == "wfa_flank":      # 2 B per wavefront offset (16-bit history) written once + pattern/text in + (n_match, span) out per job
            jobs = int(stats[14]) if int(kt["wfa_flank_rest"][1]) else int(stats[0])  # alignments of this launch
            mean_read = float(batch["read_len"].mean())
            bytes_per_launch = 2.0 * cells / max(launches, 1) + jobs * (250 + mean_read + 20)
            survey_bytes_per_launch = 4.0 * cells / max(launches, 1) + jobs * (250 + mean_read + 20)  # SURVEY 8(d) prices an offset at 4 B
        elif dom == "wfa_filter":   # history-free: sequences in, a verdict out; SURVEY 8(d) prices the DP state it keeps in registers at 4 B / offset
            jobs = int(stats[14])
            mean_read = float(batch["read_len"].mean())
            bytes_per_launch = jobs * (250 + mean_read + 20)
            survey_bytes_per_launch = 4.0 * cells / max(launches, 1) + bytes_per_launch
        elif dom == "hmm_viterbi":  # 1 B per back-pointer cell + allele in + annotation out
            bytes_per_launch = 1.0 * cells / max(launches, 1) + float(out.allele_len.sum()) * 2
        elif dom == "flank_scan":   # every read byte once + 4 B per (read, side)
            bytes_per_launch = float(batch["read_len"].sum()) + 8.0 * n_reads
        else:
            bytes_per_launch = 4.0 * cells / max(launches, 1)
        if dom not in ("wfa_flank", "wfa_filter"):
            survey_bytes_per_launch = bytes_per_launch
        traffic = None  # measured HBM bytes per launch of the same kernel / workload, when a PMC profile is committed
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("loci_per_gpu") == args.loci and dom in tj["kernels"]:
                traffic = int(tj["kernels"][dom]["bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            pass
        avg_ms = ms / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        res = {
            "metric": "loci/s", "value": round(world * args.loci * args.steps / dt, 1), "unit": "loci/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2),
            "ms_per_step_min_median_max": [round(step_ms[0], 2), round(step_ms[len(step_ms) // 2], 2), round(step_ms[-1], 2)],  # rank 0
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16+u8 (WFA), f64 (HMM)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d synthetic single-motif STR loci per GPU (motif 3-6 bp, allele <= 200 bp), "
                                   "30 reads/locus HiFi-like, 10%% truncated reads (SURVEY.md Appendix E), seed 20250509" % args.loci,
                       "loci_per_gpu": args.loci, "reads_per_locus": 30, "parallelism": "loci sharded across %d GPU(s), no collective" % world,
                       "host_threads_per_rank": host_threads},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "avg_launch_ms": round(avg_ms, 3), "launches": int(launches),
                         "algorithmic_bytes_per_launch": int(bytes_per_launch), "bytes_per_wavefront_offset": 2,
                         "frac_at_survey_4B_per_offset": round(survey_bytes_per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if avg_ms > 0 else 0.0,
                         "dp_cells_per_launch": int(cells / max(launches, 1)),
                         "dp_cells_per_s": round(cells / max(ms, 1e-9) * 1e3, 1)},
            "kernels_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in kt.items()},
            # host-visible wall time of the last step: blocked on stage A + device genotyper; consensus alignments of the loci
            # handed back to the host path; HMM enqueue / collect (its kernel overlaps the host path); host glue; whole call
            "stage_ms_last_step": {"wait_flank_location_and_genotyper": round(stats[4] / 1e6, 2), "consensus": round(stats[5] / 1e6, 2),
                                   "hmm_host_visible": round(stats[6] / 1e6, 2), "host_glue": round(stats[7] / 1e6, 2),
                                   "total": round(stats[8] / 1e6, 2)},
            "work_per_step": {"flank_wfa_jobs": int(stats[0]), "flank_wfa_jobs_first_launch": int(stats[14]), "consensus_jobs": int(stats[1]), "spanning_reads": int(stats[2]),
                              "hmm_jobs": int(stats[3]), "filter_kept": int(stats[16]), "filter_offsets": int(stats[17])},
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N = 1 only
            res["cpu_baseline"] = cpu_baseline(batch)
            nthr = min(os.cpu_count() or 1, 128)
            if nthr > 1:
                res["cpu_baseline_all_cores"] = cpu_baseline_mt(batch, nthr)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
