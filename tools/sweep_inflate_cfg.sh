#!/bin/bash
# device ingestion rate against ring size (two builds: libtrgt_hip.so = 2 KB, libtrgt_hip_r4k.so = 4 KB made with EXTRA=-DTRGT_INFL_RING=4096), waves per CU
# and caller threads.   gpurun -- bash tools/sweep_inflate_cfg.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-sweep_inflate_cfg}; mkdir -p $O; cd $R
for lib in libtrgt_hip.so libtrgt_hip_r4k.so; do
  [ -f trgt_amd/$lib ] || continue
  for w in 12 13 16; do
    if [ $lib = libtrgt_hip_r4k.so ] && [ $w = 16 ]; then continue; fi
    for callers in 1 3 5; do
      r=$(TRGT_HIP_LIB=$R/trgt_amd/$lib PROBE_WAVES=$w PROBE_DEVICE_ONLY=$callers python tools/ingest_dev_probe.py 6000 6000 1000 2>&1 | grep "device path" | tail -1)
      echo "$lib waves/CU $w: $r"
    done
  done
done 2>&1 | tee $O/summary.txt
