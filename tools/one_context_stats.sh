#!/bin/bash
# one_context_stats.sh [CONFIG]: rocprofv3 kernel stats of the one-context bench command (uncontended kernel durations), top 14 kernels
CFG=${1:-2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/oc; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python bench.py --config $CFG --steps 20 --warmup 2 --contexts 1 --no-streaming --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) | head -16 | cut -c1-60,92-150
