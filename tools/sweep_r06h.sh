#!/bin/bash
# sweep_r06h.sh TAG [FIRST]: parity sweep after the second half of round 6 (hand-written inflate loop, scalar job fields in the pre-filter, packed
# HMM back-pointer rows): catalogs of configs 2 - 5 on fresh locus ranges, the HMM row layouts, the fuzzers incl. the device ingestion.
TAG=${1:-r06h}; F=${2:-110000000}
O=gpurun_out/${TAG}_parity_sweep.txt; mkdir -p gpurun_out; : > $O
run() { echo "# parity_sweep.py $*" >> $O; python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 >> $O; }
envrun() { local e="$1"; shift; echo "# ($e) parity_sweep.py $*" >> $O; env $e python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 | sed "s|^|($e) |" >> $O; }
S=${3:-1}
run 2 $((100000 * S)) $F
run 4 $((100000 * S)) $F
run 5 $((30000 * S)) $F 2000
run 3 $((2000 * S)) $F 70
envrun "TRGT_HIP_LIB=trgt_amd/libtrgt_hip_dev.so TRGT_HMM_PPL_WIDE=1" 4 $((30000 * S)) $((F + 400000))
envrun "TRGT_HIP_LIB=trgt_amd/libtrgt_hip_dev.so TRGT_HMM_PPL_WIDE=1" 3 $((1000 * S)) $((F + 400000)) 70
envrun TRGT_HMM_NO_LONG_TB=1 3 $((1000 * S)) $((F + 500000)) 70
envrun TRGT_HMM_NO_PPL=1 4 $((30000 * S)) $((F + 600000))
python tests/tools/hmm_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/wfa_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/window_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/shortcut_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/deflate_fuzz.py 10000 12 2>&1 | grep deflate_fuzz >> $O
python tests/tools/ingest_fuzz.py 300 3 2>&1 | grep -E "RESULT|MISMATCH" >> $O
TRGT_INFLATE_COMPILER_LOOP=1 python tests/tools/ingest_fuzz.py 100 4 2>&1 | grep -E "RESULT|MISMATCH" | sed "s|^|(TRGT_INFLATE_COMPILER_LOOP=1) |" >> $O
cat $O
