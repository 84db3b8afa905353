#!/usr/bin/env python3
"""Turns a rocprofv3 results database (rocpd sqlite) into a per-kernel stats CSV:
name, calls, total_ns, avg_ns, min_ns, max_ns, percent.  Usage: rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / total, 3)])
    for r in rows[:12]:
        print(f"{r[3] / 1e3:10.1f} us avg  x{r[1]:5d}  {100.0 * r[2] / total:6.2f}%  {r[0][:110]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
