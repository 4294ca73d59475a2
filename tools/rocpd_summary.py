#!/usr/bin/env python
"""Turn a rocprofv3 results database (rocpd sqlite, what `rocprofv3 --kernel-trace --stats` leaves in this ROCm
image) into the small text summaries committed under profiles/.

  python tools/rocpd_summary.py kernels  <results.db> <out.csv>      per-kernel count / total / avg / min / max (us)
  python tools/rocpd_summary.py counters <results.db> <out.csv>      per-kernel average of every collected PMC counter
"""
import csv
import sqlite3
import sys


def kernels(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, k, t, a, mn, mx in rows:
            w.writerow([n, k, int(t), round(a, 1), round(100.0 * t / tot, 3), int(mn), int(mx)])


def counters(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                     "from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "Average", "Min", "Max"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 4), r[4], r[5]])


if __name__ == "__main__":
    {"kernels": kernels, "counters": counters}[sys.argv[1]](sys.argv[2], sys.argv[3])
