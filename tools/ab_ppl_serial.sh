for rep in 1 2; do for v in 0 1; do for c in 4 3; do
  TRGT_HMM_PPL_SERIAL=$v python bench.py --config $c --steps 40 --warmup 3 --no-streaming --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial=$v cfg$c value', d['value'], 'single', d['config']['ms_per_step_single_context'])"
done; done; done
