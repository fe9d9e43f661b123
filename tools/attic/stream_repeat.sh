#!/bin/bash
# stream_repeat.sh N [ENV=VAL ...]: the streaming legs of the default bench, N runs
n=$1; shift
for i in $(seq 1 $n); do
  env "$@" python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d["value"], d["value_streaming"], d["value_streaming_bam4"], d["value_streaming_single_context"], d["value_streaming_bam4_single_context"])' "$@"
done
