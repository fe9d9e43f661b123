#!/bin/bash
# developer helper: flank-kernel time vs persistent workgroups launched per CU
for g in "$@"; do
  TRGT_WFA_DEBUG=1 TRGT_WFA_GRID_PER_CU=$g python bench.py --steps 3 --warmup 1 2>/tmp/err.log | tail -1 > /tmp/sweep.json
  grep -m1 "\[wfa\]" /tmp/err.log
  python -c "import json; d=json.load(open('/tmp/sweep.json')); print('grid/CU', $g, d['value'], d['kernels_ms_per_step']['wfa_flank'])"
done
