#!/bin/bash
# lds_wfa_sweep.sh CONFIG "KB ...": one-context call time and consensus kernel time per LDS region size of the LDS-arena BiWFA variant
CFG=${1:-5}
cd $GRAFT_REPO_ROOT
for kb in $2; do
  echo "== TRGT_WFA_LDS_KB=$kb"
  TRGT_WFA_LDS_KB=$kb TRGT_WFA_DEBUG=1 timeout 120 python tools/timeline.py $CFG 2>&1 | grep "LDS-arena variant" | sort | uniq -c | sort -rn | head -6
  TRGT_WFA_LDS_KB=$kb timeout 200 python bench.py --config $CFG --steps 20 --warmup 2 --contexts 1 --no-streaming --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("  one-context", d["value_single_context"], "ms/step", d["ms_per_step_single_context"], d["kernels_ms_per_step"])'
done
