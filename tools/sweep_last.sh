#!/bin/bash
# sweep_last.sh TAG: the whole-catalog parity sweeps once more on the last build of round 4 (fresh locus ranges from 30.0 M (20.0 M on the first run)) + the fuzzers; summary lines into gpurun_out/TAG_parity_sweep.txt
TAG=${1:-r04last}
O=gpurun_out/${TAG}_parity_sweep.txt; mkdir -p gpurun_out; : > $O
run() { echo "# parity_sweep.py $*" >> $O; python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 >> $O; }
run 2 300000 30000000
run 4 300000 30000000
run 5 100000 30000000 2000
run 3 4000 30000000 70
run 2 60000 30400000 --bam4
run 5 20000 30400000 2000 --bam4
run 4 60000 30500000 --host-reads
run 5 20000 30500000 2000 --host-reads
run 4 40000 30600000 --rq 0.85
run 5 10000 30600000 2000 --depth 20
TRGT_HOST_CLUSTER=1 python tests/tools/parity_sweep.py 5 10000 30700000 2000 2>&1 | grep -E "RESULT|MISMATCH" | sed 's/^/(TRGT_HOST_CLUSTER=1) /' >> $O
TRGT_HMM_NO_LONG_TB=1 python tests/tools/parity_sweep.py 3 1000 30700000 70 2>&1 | grep -E "RESULT|MISMATCH" | sed 's/^/(TRGT_HMM_NO_LONG_TB=1) /' >> $O
python tests/tools/hmm_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/wfa_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/window_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/shortcut_fuzz.py 2>&1 | grep RESULT >> $O
cat $O
