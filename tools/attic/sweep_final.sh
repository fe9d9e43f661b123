O=gpurun_out/r02r_parity_sweep.txt; : > $O
run() { python tests/tools/parity_sweep.py "$@" 2>&1 | grep -E "RESULT|MISMATCH" | head -20 >> $O; }
run 2 600000 17000000
run 4 150000 17000000
run 5 60000 17000000
run 3 5600 700000 70
run 2 100000 17700000 --bam4
run 4 50000 17200000 --bam4
run 2 60000 17800000 --host-glue
run 5 20000 17100000 --depth 12
python tests/tools/hmm_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/wfa_fuzz.py 2>&1 | grep RESULT >> $O
python tests/tools/window_fuzz.py 2>&1 | grep RESULT >> $O
cat $O
