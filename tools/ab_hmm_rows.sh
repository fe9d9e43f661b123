#!/bin/bash
# A/B on one box of the position-per-lane fill's back-pointer rows: one byte per lane (default) against one byte per state
# (TRGT_HMM_PPL_WIDE=1, developer build): one-context kernel durations (rocprofv3) and call time.   gpurun -- bash tools/ab_hmm_rows.sh "4 3 2"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_hmm_rows; mkdir -p $O; cd $R
export TRGT_HIP_LIB=$R/trgt_amd/libtrgt_hip_dev.so
for cfg in ${1:-4 3 2}; do
  for wide in 0 1 0 1; do
    if [ $wide = 1 ]; then export TRGT_HMM_PPL_WIDE=1; else unset TRGT_HMM_PPL_WIDE; fi
    rm -rf $O/kt; rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python bench.py --config $cfg --steps 20 --warmup 2 --contexts 1 --no-streaming --no-cpu-baseline > $O/bench_${cfg}_$wide.json 2> $O/bench.err
    echo "== cfg$cfg wide=$wide  $(tail -1 $O/bench_${cfg}_$wide.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'])")"
    python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) | grep -i "hmm_fill\|hmm_viterbi\|traceback_long" | cut -c1-48,92-150
  done
done 2>&1 | tee $O/summary.txt
