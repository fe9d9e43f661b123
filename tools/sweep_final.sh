O=gpurun_out/r02j_parity_sweep.txt; : > $O
run() { python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 >> $O; }
run 2 600000 9000000
run 4 150000 9000000
run 5 60000 9000000
run 3 5600 500000 70
run 2 100000 9700000 --bam4
run 4 50000 9200000 --bam4
run 2 60000 9800000 --host-glue
run 5 20000 9100000 --depth 12
python tests/tools/hmm_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/wfa_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/window_fuzz.py 2>&1 | grep RESULT >> $O
cat $O
