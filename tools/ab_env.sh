# ab_env.sh VAR "v1 v2 ..." "cfgs": one-context and pool values with VAR set to each value ("-" = unset) -- same box, alternating
VAR=$1; VALS=${2:-"- 1"}; CFGS=${3:-"4 3 2 5"}
for rep in 1 2; do
for v in $VALS; do
  for c in $CFGS; do
    if [ "$v" = "-" ]; then E=""; else E="$VAR=$v"; fi
    env $E python bench.py --config $c --steps 30 --warmup 3 --no-streaming --no-cpu-baseline --no-legs --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v cfg$c value', d['value'], 'single', d['config']['value_single_context'], d['config']['ms_per_step_single_context'])"
  done
done
done
