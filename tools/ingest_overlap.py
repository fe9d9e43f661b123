"""GPU occupancy of the device ingestion with several callers: from a rocprofv3 kernel trace (rocpd database) of tools/ingest_dev_probe.py,
the union of the intervals in which any kernel ran against the span from the first to the last dispatch, per kernel name the summed time."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
name_col = "kernel_name" if "kernel_name" in scols else "display_name"
rows = c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)).fetchall()
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 400   # the last dispatches only (the timed run)
rows = rows[-tail:]
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy, cur_s, cur_e = 0, None, None
for _, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.1f ms, some kernel running %.1f ms (%.0f %%)" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
agg = {}
for nme, s, e in rows:
    a = agg.setdefault(nme.split("(")[0][-40:], [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e6
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print("  %-42s %4d launches %8.1f ms" % (k, v[0], v[1]))
