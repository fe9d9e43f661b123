#!/bin/bash
# queue_map.sh CONFIG N: kernel trace of N bench runs, to see on which hardware queue each context's kernels were dispatched
CFG=${1:-5}; N=${2:-4}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qmap; mkdir -p $O; cd $R
for i in $(seq 1 $N); do
  rocprofv3 --kernel-trace --output-format csv -d $O/k$i -o k -- python bench.py --config $CFG --no-cpu-baseline --no-streaming > $O/b$i.out 2> $O/b$i.err
  tail -1 $O/b$i.out | cut -c1-110
  f=$(find $O/k$i -name "*kernel_trace.csv" | head -1)
  python - "$f" > $O/map$i.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
m = collections.defaultdict(collections.Counter)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-40:]
    m[(r["Thread_Id"], r["Queue_Id"])][name] += 1
for (t, q), c in sorted(m.items()):
    print(t, q, dict(c.most_common(4)))
PY
  cat $O/map$i.txt | cut -c1-230
done
