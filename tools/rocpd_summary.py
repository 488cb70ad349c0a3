#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x rocpd SQLite) kernel trace: per-kernel calls / total / mean / registers.

usage: tools/rocpd_summary.py <results.db> [out.txt]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} "
             f"{'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s}"]
    for n, c, s, a, mn, mx, vg, sg, lds, scr in rows:
        n = n if len(n) <= 70 else n[:67] + "..."
        lines.append(f"{n:70s} {c:7d} {s/1e6:10.3f} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f} "
                     f"{vg or 0:5d} {sg or 0:5d} {lds or 0:7d} {scr or 0:7d}")
    lines.append(f"{'TOTAL kernel time':70s} {sum(r[1] for r in rows):7d} {tot/1e6:10.3f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
