#!/bin/bash
# developer helper: flank-kernel time vs threads per alignment (TRGT_FLANK_THREADS)
for t in "$@"; do
  TRGT_FLANK_THREADS=$t python bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 > /tmp/sweep.json
  python -c "import json; d=json.load(open('/tmp/sweep.json')); print('threads', $t, d['value'], d['kernels_ms_per_step']['wfa_flank'])"
done
