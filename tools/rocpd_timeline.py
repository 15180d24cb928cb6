#!/usr/bin/env python
"""Dispatch timeline of ONE training step from a rocprofv3 kernel trace (rocpd SQLite): kernels in start order between two
consecutive adamw_kernel dispatches, with duration and the idle gap before each; plus busy / idle totals per phase
(phases are cut at marker kernels).   python tools/rocpd_timeline.py x_results.db [step_index] [--all]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)(\w+)", n)
    if m:
        k = int(m.group(1))
        return m.group(2)[:k]
    m = re.match(r"(?:void\s+)?([\w:]+)", n)
    return (m.group(1) if m else n)[:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    step = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 5
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    s_col = "start" if "start" in cols else "start_time"
    e_col = "end" if "end" in cols else "end_time"
    rows = db.execute(f"select name, {s_col}, {e_col} from kernels order by {s_col}").fetchall()
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    a, b = marks[step] + 1, marks[step + 1] + 1
    seq = rows[a:b]
    t0 = seq[0][1]
    busy = idle = 0
    prev_end = seq[0][1]
    agg = {}
    for name, s, e in seq:
        gap = max(0, s - prev_end)
        d = e - s
        busy += d
        idle += gap
        k = short(name)
        x = agg.setdefault(k, [0, 0, 0])
        x[0] += 1; x[1] += d; x[2] += gap
        if "--all" in sys.argv:
            print(f"{(s - t0) / 1e3:10.1f} us  +{gap / 1e3:6.1f}  {d / 1e3:8.1f} us  {k}")
        prev_end = max(prev_end, e)
    print(f"step {step}: {len(seq)} dispatches, wall {(seq[-1][2] - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle gaps {idle / 1e6:.3f} ms")
    print(f"{'kernel':40s} {'n':>5s} {'busy ms':>9s} {'gap-before ms':>14s}")
    for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f"{k:40s} {n:5d} {d / 1e6:9.3f} {g / 1e6:14.3f}")


if __name__ == "__main__":
    main()
