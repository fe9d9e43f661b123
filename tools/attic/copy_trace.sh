#!/bin/bash
# copy_trace.sh: which engine moves what in the submit / wait pipeline (4-bit reads in pinned memory, one context): kernel trace + memory-copy trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/copytrace; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/k -o k -- python tools/timeline_stream.py 6 > $O/t.out 2> $O/t.err
ls $O/k
python - $O/k <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
for f in glob.glob(d + "/*memory_copy_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    print("memory copies:", len(rows), "columns", list(rows[0].keys()) if rows else None)
    agg = collections.defaultdict(lambda: [0, 0, 0.0])
    for r in rows:
        k = (r.get("Direction"), )
        b = int(r.get("Bytes", r.get("Size", 0)) or 0)
        agg[k][0] += 1; agg[k][1] += b; agg[k][2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, v in agg.items(): print("  ", k, "n=%d bytes=%.1f MB total %.2f ms" % (v[0], v[1] / 1e6, v[2]))
    big = [r for r in rows if int(r.get("Bytes", r.get("Size", 0)) or 0) > (1 << 20)]
    for r in big[-8:]: print("   big:", r.get("Direction"), int(r.get("Bytes", r.get("Size", 0))) >> 20, "MB", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms")
for f in glob.glob(d + "/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in rows:
        n = r["Kernel_Name"].split("(")[0][-40:]
        if "copy" in n.lower() or "fill" in n.lower():
            agg[n][0] += 1; agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; agg[n][2] = max(agg[n][2], int(r["Grid_Size_X"]))
    for k, v in agg.items(): print("  kernel", k, "n=%d total %.2f ms max grid %d" % tuple(v))
PY
