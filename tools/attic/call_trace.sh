#!/bin/bash
# call_trace.sh N: N runs of tools/timeline_timing.py (config 5, one context, 5 calls) under the kernel trace; for the last call of
# each run: the host timeline's end and every kernel of >= 0.15 ms with its start, duration, stream and hardware queue
N=${1:-3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/calltrace; mkdir -p $O; cd $R
for i in $(seq 1 $N); do
  rocprofv3 --kernel-trace --output-format csv -d $O/k$i -o k -- python tools/timeline_timing.py > $O/t$i.out 2> $O/t$i.err
  echo "run $i: $(grep 'all collected' $O/t$i.err | tail -1)"
  f=$(find $O/k$i -name "*kernel_trace.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) > t_end - 52e6]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d >= 0.15:
        print("  %8.3f %8.3f  %-44s stream %2s queue %s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, r["Kernel_Name"].split("(")[0][-44:], r["Stream_Id"], r["Queue_Id"], r["Grid_Size_X"]))
PY
done
