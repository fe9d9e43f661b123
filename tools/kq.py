#!/usr/bin/env python3
"""developer helper: one-line summary of a bench.py JSON line read from stdin."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("kernels_ms_per_step"), d.get("stage_ms_last_step"))
