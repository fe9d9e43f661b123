#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd SQLite output) per kernel: calls, total / avg / min / max ms.

usage: tools/rocprof_summary.py gpurun_out/prof_xx/bench_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = c.execute("select s.%s, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id"
                     % (name_col, kd, ks)).fetchall() if "kernel_id" in cols else []
    agg = {}
    for name, st, en, gx, wx in rows:
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0, gx, wx])
        d = (en - st) / 1e6
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1.0
    print("%-90s %6s %12s %10s %10s %10s %6s %10s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct", "grid_x", "wg_x"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-90s %6d %12.3f %10.3f %10.3f %10.3f %6.1f %10d %6d" % (name[:90], a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot, a[4], a[5]))


if __name__ == "__main__":
    main(sys.argv[1])
