for v in 0 8 10; do
  for c in 2 4; do
    TRGT_FILTER_PER_CU=$v python bench.py --config $c --steps 30 --warmup 3 --no-streaming --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('per_cu=$v cfg$c value', d['value'], 'single', d['config']['value_single_context'], d['config']['ms_per_step_single_context'])"
  done
done
