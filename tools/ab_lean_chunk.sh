for ch in 1 2 3 4; do
  TRGT_LEAN_CHUNK=$ch python bench.py --config 5 --steps 40 --warmup 3 --no-streaming --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk=$ch cfg5 value', d['value'], 'single', d['config']['ms_per_step_single_context'])"
done
