#!/bin/bash
# call_trace3.sh CONFIG [WINDOW_MS]: every kernel / copy / fill dispatch of the last call of tools/timeline.py CONFIG in start order
# (start, duration, gap to the previous end on the same stream, name, stream) + the host timeline of that call
CFG=${1:-2}; WIN=${2:-9}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/calltrace3; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/k -o k -- python tools/timeline.py $CFG > $O/t.out 2> $O/t.err
awk '/---- call 2/{p=1} p' $O/t.err > $O/host_timeline.txt
python - "$(find $O/k -name '*kernel_trace.csv' | head -1)" $WIN > $O/dispatches.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); win = float(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) > t_end - win * 1e6]
t0 = int(sel[0]["Start_Timestamp"])
last = {}
print("n dispatches in window:", len(sel))
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    st = r["Stream_Id"]
    gap = (s - last[st]) / 1e3 if st in last else 0.0
    last[st] = e
    print("%8.3f %8.3f  gap %7.1f us  %-48s stream %2s q %s grid %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap, r["Kernel_Name"].split("(")[0][-48:], st, r["Queue_Id"], r["Grid_Size_X"]))
PY
tail -40 $O/host_timeline.txt; head -120 $O/dispatches.txt
