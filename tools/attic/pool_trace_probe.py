"""developer probe: per-context call durations and gaps of trgt_locus_batch_many (TRGT_POOL_TRACE) for 4-bit reads resident in HBM vs
in pinned host memory: pool_trace_probe.py [contexts] [batches]"""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from trgt_amd import locus, synth, _lib
    mode, K, N = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    b = synth.generate(10000, first_locus=0, config=2)
    params = locus.Params(host_threads=8)
    fd = torch.from_numpy(b["flank_blob"]).cuda()
    pk = locus.pack_bam4(b, pinned=True)
    kw = dict(reads_dev=torch.from_numpy(pk["read_blob"]).cuda()) if mode == "resident" else {}
    pool = _lib.Pool([0] * K)
    outs = [locus.BatchOutputs(b) for _ in range(K)]
    locus.run_many(pool, [pk] * (3 * K), params, outs, flank_dev=fd, out_per_context=True, **kw)
    torch.cuda.synchronize()
    print("==== measured", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    locus.run_many(pool, [pk] * N, params, outs, flank_dev=fd, out_per_context=True, **kw)
    torch.cuda.synchronize()
    print("RATE %s %.3f" % (mode, 10000 * N / (time.perf_counter() - t0) / 1e6))
    sys.exit(0)
K = sys.argv[1] if len(sys.argv) > 1 else "4"
N = sys.argv[2] if len(sys.argv) > 2 else "48"
for mode in ("resident", "pinned"):
    r = subprocess.run([sys.executable, __file__, "child", mode, K, N], env=dict(os.environ, TRGT_POOL_TRACE="1", TRGT_TIMELINE="1"), capture_output=True, text=True)
    err = r.stderr.split("==== measured")[-1]
    rows = [l.split() for l in err.splitlines() if l.startswith("[pool]")]
    by = {}
    for _, w, bidx, s, e in rows:
        by.setdefault(int(w), []).append((float(s), float(e)))
    print(r.stdout.strip().splitlines()[-1])
    import re, collections
    marks = collections.defaultdict(list)
    for l in err.splitlines():
        m = re.match(r"\[tl\] (.*?)\s+([\d.]+) ms  ctx=(\w+)", l)
        if m: marks[m.group(1).strip()].append(float(m.group(2)))
    for k in ("set-up", "stage A enqueued", "hmm1 enqueued (device-resolved job list)", "evA", "stream2 synced", "stageB", "hmm2 enqueued", "hmm1 collected", "all collected"):
        v = marks.get(k, [])
        if v: print("    %-42s mean %6.2f  min %6.2f  max %6.2f  (n=%d)" % (k, sum(v) / len(v), min(v), max(v), len(v)))
    for w, v in sorted(by.items()):
        v.sort()
        dur = [e - s for s, e in v]; gap = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
        print("     durations:", " ".join("%.1f" % d for d in dur))
        print("  ctx %d: %d calls, wait() %.2f ms mean (min %.2f max %.2f), between waits %.3f ms mean (max %.3f)" % (w, len(v), sum(dur) / len(dur), min(dur), max(dur), sum(gap) / max(1, len(gap)), max(gap) if gap else 0))
