#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel (usage: pmc_summary.py file.csv [...])."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add((path, r["Dispatch_Id"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-50s dispatches=%d" % (k, len(disp[k])))
    for c, x in sorted(v.items()):
        print("    %-28s %16.0f" % (c, x))
