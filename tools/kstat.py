#!/usr/bin/env python3
"""avg / min / max ms of the kernels whose name contains one of the given words, from a rocprofv3 kernel_stats.csv:  python tools/kstat.py stats.csv word ..."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(w in r["Name"] for w in sys.argv[2:]):
        print("%-44s calls %4s  avg %.3f  min %.3f  max %.3f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
