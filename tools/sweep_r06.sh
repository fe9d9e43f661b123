#!/bin/bash
# sweep_r06.sh TAG [FIRST]: whole-catalog parity sweeps of round 6 (fresh locus ranges from FIRST, default 100.0 M) + the fuzzers, incl. the
# device-ingestion fuzz; summary lines into gpurun_out/TAG_parity_sweep.txt.  Switches the release library does not read any more
# (TRGT_HMM_PPL_PER_CLASS: a settled A/B) run on the developer build (TRGT_HIP_LIB=trgt_amd/libtrgt_hip_dev.so).
TAG=${1:-r06}; F=${2:-100000000}
O=gpurun_out/${TAG}_parity_sweep.txt; mkdir -p gpurun_out; : > $O
run() { echo "# parity_sweep.py $*" >> $O; python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 >> $O; }
envrun() { local e="$1"; shift; echo "# ($e) parity_sweep.py $*" >> $O; env $e python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 | sed "s|^|($e) |" >> $O; }
run 2 300000 $F
run 4 300000 $F
run 5 100000 $F 2000
run 3 4000 $F 70
run 2 60000 $((F + 400000)) --bam4
run 5 20000 $((F + 400000)) 2000 --bam4
run 4 60000 $((F + 500000)) --host-reads
run 5 20000 $((F + 500000)) 2000 --host-reads
run 4 40000 $((F + 600000)) --rq 0.85
run 5 10000 $((F + 600000)) 2000 --depth 20
envrun TRGT_HOST_CLUSTER=1 5 10000 $((F + 700000)) 2000
envrun TRGT_HMM_NO_LONG_TB=1 3 1000 $((F + 700000)) 70
envrun TRGT_HMM_NO_PPL=1 4 40000 $((F + 800000))
envrun TRGT_HMM_NO_PPL=1 3 1000 $((F + 800000)) 70
envrun TRGT_NO_ZERO_ARENA=1 4 40000 $((F + 900000))
envrun TRGT_NO_ZERO_ARENA=1 5 10000 $((F + 900000)) 2000
envrun "TRGT_HIP_LIB=trgt_amd/libtrgt_hip_dev.so TRGT_HMM_PPL_PER_CLASS=1" 4 40000 $((F + 1000000))
envrun "TRGT_HIP_LIB=trgt_amd/libtrgt_hip_dev.so TRGT_HMM_PPL_PER_CLASS=1" 3 1000 $((F + 1000000)) 70
python tests/tools/hmm_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/wfa_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/window_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/shortcut_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/deflate_fuzz.py 10000 11 2>&1 | grep deflate_fuzz >> $O
python tests/tools/ingest_fuzz.py 300 1 2>&1 | grep -E "RESULT|MISMATCH" >> $O
python tests/tools/ingest_fuzz.py 300 2 2>&1 | grep -E "RESULT|MISMATCH" >> $O
cat $O
