#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel (usage: pmc_summary.py file.csv [...]).
Per counter: the total over its dispatches, how many dispatches carried it (a counter sits in ONE of the runs), and the average per
dispatch -- the figure to quote.  (Until round 4 this printed one dispatch count per kernel, the dispatches of all runs together, and
figures "per dispatch" derived from it were too small by the number of runs sharing the file list.)"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
rows = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        rows[k][r["Counter_Name"]] += 1
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-50s" % k)
    for c, x in sorted(v.items()):
        n = rows[k][c]
        print("    %-28s %16.0f  dispatches=%-5d per dispatch %14.0f" % (c, x, n, x / max(1, n)))
