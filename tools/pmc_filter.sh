#!/bin/bash
# SQ counters of the flank pre-filter kernel only (run through gpurun): bash tools/pmc_filter.sh <tag>
TAG=${1:-flt}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "wfa_filter" -d $O/p$i -o p$i --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-legs --no-e2e --no-streaming > $O/p$i.log 2>&1
done
python tools/pmc_summary.py $(find $O -name "*counter_collection.csv") > $O/${TAG}_sq.txt
cat $O/${TAG}_sq.txt
