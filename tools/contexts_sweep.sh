#!/bin/bash
# contexts_sweep.sh CONFIG "K1 K2 ...": value of bench.py for several pool sizes
CFG=${1:-2}; shift
for k in ${1:-"2 3 4 5 6"}; do
  python bench.py --config $CFG --contexts $k --no-cpu-baseline --no-streaming 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])' $k
done
