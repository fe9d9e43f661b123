# Round-3 parity sweep on the GPU box: tests/tools/parity_sweep.py over fresh locus ranges in the modes of the call + the three fuzzers.
# Usage (through gpurun): bash tools/sweep_r03.sh <tag> [scale] [first-locus shift]   -- scale 1 = 0.9 M loci (about 8 minutes); a shift moves every range to fresh loci
TAG=${1:-r03}; SC=${2:-1}; SH=${3:-0}
O=gpurun_out/${TAG}_parity_sweep.txt; : > $O
run() { echo "== parity_sweep.py $*" >> $O; python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 >> $O; }
run 2 $((400000*SC)) $((31000000+SH))
run 4 $((150000*SC)) $((31000000+SH))
run 5 $((60000*SC)) $((31000000+SH))
run 3 5600 $((1400000+SH)) 70
run 2 $((100000*SC)) $((31700000+SH)) --bam4
run 2 $((60000*SC)) $((31800000+SH)) --host-reads
run 4 $((50000*SC)) $((31200000+SH)) --bam4
run 2 $((60000*SC)) $((31900000+SH)) --host-glue
run 5 $((20000*SC)) $((31100000+SH)) --depth 12
TRGT_HOST_REPAIR=1 run 2 $((60000*SC)) $((32000000+SH))
TRGT_REPAIR_MAX_SEG=60 run 4 $((50000*SC)) $((32100000+SH))
echo "== fuzzers" >> $O
python tests/tools/hmm_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/wfa_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/window_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/shortcut_fuzz.py 4000 $((30*SC)) 1 2>&1 | grep RESULT >> $O
cat $O
