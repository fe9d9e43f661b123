#!/bin/bash
# developer helper: whole-step time vs chunk lanes
for l in 1 2; do for sa in 0 1; do
  if [ $sa = 1 ]; then export TRGT_SERIAL_A=1; else unset TRGT_SERIAL_A; fi
  TRGT_LOCUS_LANES=$l python bench.py --steps 5 --warmup 2 2>/dev/null | tail -1 > /tmp/sweep.json
  python -c "import json; d=json.load(open('/tmp/sweep.json')); print('lanes', $l, 'serialA', $sa, d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"
done; done
