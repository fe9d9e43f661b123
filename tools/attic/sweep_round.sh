#!/bin/bash
# sweep_round.sh TAG: the whole-catalog parity sweeps of a round (fresh locus ranges) + the fuzzers, summary lines into gpurun_out/TAG_parity_sweep.txt
TAG=${1:-rXX}
O=gpurun_out/${TAG}_parity_sweep.txt; mkdir -p gpurun_out; : > $O
run() { python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 >> $O; }
run 2 300000 7000000
run 4 100000 7000000
run 5 40000 7000000
run 2 60000 7400000 --bam4
run 5 20000 7400000 --bam4
run 2 60000 7500000 --host-reads
run 4 40000 7500000 --rq 0.85
python tests/tools/hmm_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/wfa_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/window_fuzz.py 2>&1 | grep RESULT >> $O
cat $O
