#!/bin/bash
# PMC passes over the device ingestion probe (tools/ingest_dev_probe.py, device path only): what the inflate kernel's waves spend their cycles on.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_inflate}; mkdir -p $O; cd $R
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1))
  PROBE_DEVICE_ONLY=1 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i --output-format csv -- python tools/ingest_dev_probe.py 1000 6000 1000 > $O/p$i.log 2>&1
done
python tools/pmc_summary.py $(find $O -name "*counter_collection.csv") > $O/summary.txt
grep -i -A12 "inflate" $O/summary.txt | head -60
