#!/bin/bash
# call_trace2.sh CONFIG [WINDOW_MS] [MIN_MS] [N_LOCI]: kernel trace of tools/timeline.py CONFIG; kernels (>= 0.15 ms) of the last WINDOW_MS of the run with start, duration, stream, queue
CFG=${1:-3}; WIN=${2:-70}; MIN=${3:-0.15}; NL=${4:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/calltrace2; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/k -o k -- python tools/timeline.py $CFG $NL > $O/t.out 2> $O/t.err
grep -E "evA|stageB|hmm2 enq|collected" $O/t.err | tail -5
python - "$(find $O/k -name '*kernel_trace.csv' | head -1)" $WIN $MIN <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); win = float(sys.argv[2]); dmin = float(sys.argv[3])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) > t_end - win * 1e6]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d >= dmin:
        print("  %8.3f %8.3f  %-44s stream %2s queue %s grid %s wg %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, r["Kernel_Name"].split("(")[0][-44:], r["Stream_Id"], r["Queue_Id"], r["Grid_Size_X"], r["Workgroup_Size_X"]))
PY
