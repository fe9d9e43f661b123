#!/bin/bash
# inflate kernel time of a 1 000-locus chunk against the waves (= BGZF blocks in flight) per CU.   gpurun -- bash tools/sweep_inflate_waves.sh [tag] "W1 W2 ..."
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-sweep_waves}; mkdir -p $O; cd $R
for w in ${2:-4 6 8 9 10 12 13}; do
  PROBE_WAVES=$w PROBE_DEVICE_ONLY=1 rocprofv3 --kernel-trace --stats -d $O/w$w -o kt --output-format csv -- python tools/ingest_dev_probe.py 2000 6000 1000 > $O/w$w.log 2>&1
  echo "waves/CU $w: $(python tools/kstat.py $O/w$w/kt_kernel_stats.csv inflate)  $(grep 'device path' $O/w$w.log | tail -1)"
done 2>&1 | tee $O/summary.txt
