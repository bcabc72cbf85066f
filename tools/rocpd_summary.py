#!/usr/bin/env python
"""Summarise rocprofv3 rocpd SQLite outputs (kernel stats + PMC counters) as text.

    python tools/rocpd_summary.py gpurun_out/prof_TAG  > profiles/TAG_summary.txt
"""
import glob
import os
import sqlite3
import sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
        "group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["%-64s %7s %14s %12s %12s %12s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%")]
    for r in rows:
        out.append("%-64s %7d %14d %12.0f %12d %12d %6.2f" % (r[0][:64], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total))
    return "\n".join(out)


def counters(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = cur.execute(
        f"select {name_col}, counter_name, count(*), avg(value), sum(value) from counters_collection "
        f"group by {name_col}, counter_name order by 1, 2").fetchall()
    out = ["%-44s %-26s %6s %20s" % ("kernel", "counter", "n", "avg per dispatch")]
    for r in rows:
        out.append("%-44s %-26s %6d %20.1f" % (r[0][:44], r[1], r[2], r[3]))
    return "\n".join(out)


def main():
    root = sys.argv[1]
    for d in sorted(os.listdir(root)):
        for db in glob.glob(os.path.join(root, d, "*.db")):
            print(f"==== {d} ({os.path.basename(db)})")
            try:
                if d.startswith("trace"):
                    print(kernel_stats(db))
                else:
                    print(counters(db))
            except Exception as exc:  # keep going: schema differs between rocprofv3 builds
                print("  failed:", exc)
            print()


if __name__ == "__main__":
    main()
