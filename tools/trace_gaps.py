#!/usr/bin/env python
"""Timeline of a rocprofv3 --kernel-trace database: per kernel name the busy time, and the idle gaps between consecutive
dispatches (the launch-bound part of a chain of short kernels).   python tools/trace_gaps.py gpurun_out/prof_X/trace"""
import glob
import os
import sqlite3
import sys

dbs = glob.glob(os.path.join(sys.argv[1], "*.db"))
if not dbs:
    raise SystemExit("no trace database")
cur = sqlite3.connect(dbs[0]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
if not rows:
    raise SystemExit("no kernels")
print("==== timeline (%d dispatches)" % len(rows))
span = rows[-1][2] - rows[0][1]
busy = sum(e - s for _, s, e in rows)
gaps = [max(0, rows[i + 1][1] - rows[i][2]) for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 200_000]  # gaps above 0.2 ms are host pauses (set-up, synchronisation), not launch gaps
print("first start -> last end %.3f ms, kernels busy %.3f ms, launch gaps < 0.2 ms: %d totalling %.3f ms (mean %.1f us)" % (
    span / 1e6, busy / 1e6, len(small), sum(small) / 1e6, (sum(small) / max(1, len(small))) / 1e3))
# the steady part: the last 60 % of the dispatches
tail = rows[int(len(rows) * 0.4):]
tspan = tail[-1][2] - tail[0][1]
tbusy = sum(e - s for _, s, e in tail)
print("last 60 %% of the dispatches: span %.3f ms, busy %.3f ms (%.1f %%)" % (tspan / 1e6, tbusy / 1e6, 100.0 * tbusy / max(1, tspan)))
by = {}
for (n, s, e), g in zip(rows[1:], gaps):
    d = by.setdefault(n.split("(")[0][:60], [0, 0, 0])
    d[0] += 1
    d[1] += e - s
    d[2] += g if g < 200_000 else 0
print("%-62s %7s %12s %14s" % ("kernel", "calls", "avg_us", "avg gap before_us"))
for n, (c, t, g) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %7d %12.2f %14.2f" % (n, c, t / c / 1e3, g / c / 1e3))
