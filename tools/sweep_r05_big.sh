#!/bin/bash
# sweep_r05_big.sh TAG: 3 M loci of configs 2 / 4 (1.2 M each), 5 (0.4 M) and 3 (12 k) at fresh locus ranges (50.0 M), default modes only: the volume run for
# the position-per-lane HMM fill and the zero arena (the mode matrix is tools/sweep_r05.sh)
TAG=${1:-r05big}; F=${2:-50000000}
O=gpurun_out/${TAG}_parity_sweep.txt; mkdir -p gpurun_out; : > $O
run() { echo "# parity_sweep.py $*" >> $O; python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 >> $O; }
run 2 1200000 $F
run 4 1200000 $F
run 5 400000 $F 2000
run 3 12000 $F 70
cat $O
