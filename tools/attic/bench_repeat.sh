#!/bin/bash
# bench_repeat.sh CONFIG N [ENV=VAL ...]: run the default bench N times, print value / ms_per_step / single-context value of each run
cfg=$1; n=$2; shift 2
for i in $(seq 1 $n); do
  env "$@" python bench.py --config $cfg --no-cpu-baseline --no-streaming 2>/dev/null | tail -1 | \
    python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d["value"], d["ms_per_step"], d["value_single_context"])' "$@"
done
