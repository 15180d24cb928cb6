#!/usr/bin/env python
"""Kernel statistics (the table `rocprofv3 --kernel-trace --stats` prints) from the rocpd SQLite database that
rocprofv3 writes on this image: one CSV row per kernel — calls, total / average / min / max duration in ns, share.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, n, tot, avg, mn, mx in rows:
        w.writerow([name, n, tot, f"{avg:.1f}", f"{100.0 * tot / total:.2f}", mn, mx])


if __name__ == "__main__":
    main(sys.argv[1])
