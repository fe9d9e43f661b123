# pmc_probe.sh: SQ counters of tools/biwfa_probe.py (uniform consensus-like BiWFA batches), per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_probe
rm -rf $O; mkdir -p $O
cd $R
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU" "SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_WAVE32"; do
  i=$((i+1))
  timeout 100 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i --output-format csv -- python tools/biwfa_probe.py 60000 200 > $O/p$i.log 2>&1
done
python tools/pmc_summary.py $(find $O -name "*counter_collection.csv") > $O/summary.txt
grep -A24 "lean::wfa_lean_kernel<3, 1" $O/summary.txt | head -30
