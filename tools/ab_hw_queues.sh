# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4 per device and process?); streams that share a queue run one after the other.
# One-context and pool values by that setting -- same box
for q in ${QS:-"" 8 16 24}; do
  for c in 4 3 2 5; do
    env ${q:+GPU_MAX_HW_QUEUES=$q} python bench.py --config $c --steps 30 --warmup 3 --no-streaming --no-cpu-baseline --no-legs --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues=${q:-default} cfg$c value', d['value'], 'single', d['config']['value_single_context'], d['config']['ms_per_step_single_context'])"
  done
done
