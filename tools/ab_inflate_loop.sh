#!/bin/bash
# A/B of the inflate kernel's symbol loop on ONE box: the hand-written loop (default) against the compiler's (TRGT_INFLATE_COMPILER_LOOP=1):
# device ingestion probe (one caller; three callers), rocprofv3 kernel stats of each.   gpurun -- bash tools/ab_inflate_loop.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ab_inflate}; mkdir -p $O; cd $R
for mode in asm compiler; do
  unset TRGT_INFLATE_COMPILER_LOOP
  if [ $mode = compiler ]; then export TRGT_INFLATE_COMPILER_LOOP=1; fi
  PROBE_DEVICE_ONLY=1 rocprofv3 --kernel-trace --stats -d $O/kt_$mode -o kt --output-format csv -- python tools/ingest_dev_probe.py 2000 6000 1000 > $O/probe1_$mode.log 2>&1
  PROBE_DEVICE_ONLY=3 python tools/ingest_dev_probe.py 4000 6000 1000 > $O/probe3_$mode.log 2>&1
  f=$(find $O/kt_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode" ; grep -h "device path" $O/probe1_$mode.log $O/probe3_$mode.log | tail -2; python tools/kstat.py $f inflate
done 2>&1 | tee $O/summary.txt
