for rep in 1 2; do
for lib in new prev; do
  for c in ${CFGS:-4 5}; do
    if [ $lib = prev ]; then export TRGT_HIP_LIB=$PWD/trgt_amd/libtrgt_hip_prev.so; else unset TRGT_HIP_LIB; fi
    python bench.py --config $c --steps 40 --warmup 3 --no-streaming --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib cfg$c value', d['value'], 'single', d['config']['ms_per_step_single_context'])"
  done
done
done
