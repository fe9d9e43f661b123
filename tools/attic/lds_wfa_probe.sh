#!/bin/bash
# lds_wfa_probe.sh CONFIG: how many BiWFA jobs fit the LDS-arena variant (TRGT_WFA_DEBUG) + kernel stats of the one-context bench
CFG=${1:-5}
cd $GRAFT_REPO_ROOT
TRGT_WFA_DEBUG=1 timeout 120 python tools/timeline.py $CFG 2>&1 | grep "LDS-arena variant: metric" | sort | uniq -c | sort -rn | head -12
timeout 200 bash tools/one_context_stats.sh $CFG
