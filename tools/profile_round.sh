#!/bin/bash
# Collect the per-round profile set on the GPU box (run through gpurun): kernel-trace stats of a bench run, then PMC passes
# (each in its own run, --kernel-trace only, as the pool requires).  Usage: bash tools/profile_round.sh <tag> [config] [pmc: 0/1]
TAG=${1:-rXX}
CFG=${2:-2}
PMC=${3:-1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -o bench -- python bench.py --config $CFG --steps 20 --warmup 2 > $O/bench.json 2> $O/bench.err
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/${TAG}_kernel_stats.txt
tail -1 $O/bench.json > $O/${TAG}_bench.json
head -14 $O/${TAG}_kernel_stats.txt
# the same workload through one context only (no pool, no streaming leg): the launches of this trace are the uncontended ones the
# roofline block of bench.py is computed from (its single-context loop), so the per-kernel averages of the two must agree
rocprofv3 --kernel-trace --stats -d $O/kt1 -o bench1 -- python bench.py --config $CFG --steps 20 --warmup 2 --contexts 1 --no-streaming --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err
python tools/rocprof_summary.py $(find $O/kt1 -name "*.db" | head -1) > $O/${TAG}_kernel_stats_one_context.txt
tail -1 $O/bench1.json > $O/${TAG}_bench_one_context.json
head -8 $O/${TAG}_kernel_stats_one_context.txt
if [ "$PMC" = "1" ]; then
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i --output-format csv -- python bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-streaming > $O/p$i.log 2>&1
done
python tools/pmc_summary.py $O/p1/*counter_collection.csv $O/p2/*counter_collection.csv > $O/${TAG}_pmc_hbm_fetch_write.txt
python tools/pmc_summary.py $O/p3/*counter_collection.csv $O/p4/*counter_collection.csv $O/p5/*counter_collection.csv $O/p6/*counter_collection.csv > $O/${TAG}_pmc_sq_counters.txt
head -12 $O/${TAG}_pmc_hbm_fetch_write.txt
fi
