#!/usr/bin/env python3
"""Timeline of one frame out of a rocprofv3 (rocpd SQLite) kernel trace: every kernel between two launches of the frame's
first kernel, with its start offset, duration and the idle gap before it.

usage: tools/rocpd_timeline.py <results.db> <first-kernel-substring> [occurrence (default: the middle one)]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    first = sys.argv[2]
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    if len(starts) < 3:
        print("kernel not found often enough:", first, len(starts))
        return
    k = int(sys.argv[3]) if len(sys.argv) > 3 else len(starts) // 2
    a, b = starts[k], starts[k + 1]
    t0 = rows[a][1]
    prev_end = rows[a][1]
    busy = 0
    print(f"{'kernel':58s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>8s}")
    for name, s, e in rows[a:b]:
        print(f"{name[:58]:58s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f}")
        busy += e - s
        prev_end = max(prev_end, e)
    span = rows[b][1] - t0
    print(f"frame: {b - a} kernels, span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
