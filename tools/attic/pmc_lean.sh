# pmc_lean.sh [CONFIG]: SQ instruction counters of the one-context bench command, per kernel (what the lean alignment kernels issue)
CFG=${1:-5}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lean
rm -rf $O; mkdir -p $O
cd $R
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i --output-format csv -- python bench.py --config $CFG --steps 1 --warmup 1 --contexts 1 --no-streaming --no-cpu-baseline --no-e2e > $O/p$i.log 2>&1
done
python tools/pmc_summary.py $(find $O -name "*counter_collection.csv") > $O/summary.txt
grep -i "lean\|kernel" $O/summary.txt | head -30
