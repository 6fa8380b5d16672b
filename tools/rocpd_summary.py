#!/usr/bin/env python3
"""Summarises rocprofv3's rocpd .db outputs (kernel-trace stats and PMC passes) as text for profiles/."""
import glob
import os
import sqlite3
import sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["%-60s %6s %14s %14s %14s %14s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for r in rows:
        out.append("%-60s %6d %14d %14.0f %14d %14d %6.2f%%" % (r[0][:60], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total))
    return "\n".join(out)


def pmc(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute("select %s, counter_name, count(*), sum(value), avg(value) from counters_collection group by 1, 2 order by 1, 2" % name_col).fetchall()
    out = ["%-44s %-28s %6s %22s %22s" % ("kernel", "counter", "disp", "sum", "avg/dispatch")]
    for r in rows:
        out.append("%-44s %-28s %6d %22.0f %22.1f" % (r[0][:44], r[1], r[2], r[3], r[4]))
    return "\n".join(out)


if __name__ == "__main__":
    d = sys.argv[1]
    for db in sorted(glob.glob(os.path.join(d, "*", "*_results.db"))):
        print("== " + os.path.relpath(db, d))
        try:
            print(kernel_stats(db) if "stats" in os.path.basename(db) else pmc(db))
        except Exception as e:  # schema differences between rocprofv3 versions
            print("could not summarise: %r" % (e,))
        print()
