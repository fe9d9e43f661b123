#!/bin/bash
# sweep_band.sh TAG: parity sweeps after the banded back-trace of the kept flank alignments went in (fresh locus ranges from 10.0 M; configs 2 and 4 are
# the ones whose flank location meets the pre-filter) + the flank fuzzers; summary lines into gpurun_out/TAG_parity_sweep.txt
TAG=${1:-r04band}
O=gpurun_out/${TAG}_parity_sweep.txt; mkdir -p gpurun_out; : > $O
run() { echo "# parity_sweep.py $*" >> $O; python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 >> $O; }
run 2 300000 10000000
run 4 300000 10000000
run 5 60000 10000000 2000
run 2 60000 10400000 --bam4
run 4 60000 10500000 --host-reads
run 4 40000 10600000 --rq 0.85
TRGT_HEAVY_BAND=256 python tests/tools/parity_sweep.py 2 100000 10700000 2>&1 | grep -E "RESULT|MISMATCH" | sed 's/^/(TRGT_HEAVY_BAND=256) /' >> $O
TRGT_HEAVY_BAND=32 python tests/tools/parity_sweep.py 4 100000 10700000 2>&1 | grep -E "RESULT|MISMATCH" | sed 's/^/(TRGT_HEAVY_BAND=32) /' >> $O
python tests/tools/wfa_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/window_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/shortcut_fuzz.py 2>&1 | grep RESULT >> $O
cat $O
