#!/bin/bash
# scratch_trace.sh CONFIG N: rocprofv3 scratch-memory trace of N bench runs (does the runtime allocate/reclaim scratch per dispatch?)
CFG=${1:-5}; N=${2:-3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/scratch; mkdir -p $O; cd $R
for i in $(seq 1 $N); do
  rocprofv3 --scratch-memory-trace --output-format csv -d $O/s$i -o s -- python bench.py --config $CFG --no-cpu-baseline --no-streaming > $O/b$i.out 2> $O/b$i.err
  tail -1 $O/b$i.out | cut -c1-110
  f=$(find $O/s$i -name "*scratch_memory*.csv" | head -1)
  echo "scratch events: $(wc -l < $f)"; head -5 $f
done
