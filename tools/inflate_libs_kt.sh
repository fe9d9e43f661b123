#!/bin/bash
# inflate kernel time (one caller, rocprofv3) of several builds of the library.   gpurun -- bash tools/exp_nowait.sh "libA.so libB.so"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp_libs; mkdir -p $O; cd $R
for lib in ${1:-libtrgt_hip.so libtrgt_hip_r4k.so}; do
  TRGT_HIP_LIB=$R/trgt_amd/$lib PROBE_DEVICE_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/$lib -o kt --output-format csv -- python tools/ingest_dev_probe.py 2000 6000 1000 > $O/$lib.log 2>&1
  echo "$lib: $(python tools/kstat.py $O/$lib/kt_kernel_stats.csv inflate)  $(grep 'device path' $O/$lib.log | tail -1)"
done
