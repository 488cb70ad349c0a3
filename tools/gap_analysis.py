#!/usr/bin/env python3
"""Idle device time between the kernels of a step, from a rocprofv3 kernel trace (rocpd SQLite).

usage: tools/gap_analysis.py <results.db> [steps=20] [anchor=k_stage_a_fused]

A step = from one launch of the anchor kernel to the next.  For the last `steps` of them: the union of the kernels' busy intervals,
the idle remainder, and the idle time attributed to (kernel before the gap -> kernel after it), averaged per step.
"""
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = n.replace("edgehip::", "").replace("void ", "")
    for cut in ("<", "("):
        if cut in n:
            n = n[:n.index(cut)]
    return n


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    anchor = sys.argv[3] if len(sys.argv) > 3 else "k_stage_a_fused"
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = db.execute(f'select name, "{s}", "{e}" from kernels order by "{s}"').fetchall()
    marks = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(marks) < steps + 1:
        sys.exit(f"only {len(marks)} launches of {anchor}")
    # the timed steps run back to back: keep the `steps` shortest anchor-to-anchor intervals (the profiled steps behind them and the
    # checks between the legs are longer)
    print("anchor-to-anchor intervals in launch order, us:", " ".join(f"{(rows[b][1] - rows[a][1]) / 1e3:.0f}" for a, b in zip(marks[:-1], marks[1:])))
    spans = sorted((rows[b][1] - rows[a][1], a, b) for a, b in zip(marks[:-1], marks[1:]))[:steps]
    pairs = sorted((a, b) for _, a, b in spans)
    gaps = defaultdict(float)
    count = defaultdict(int)
    busy = idle = 0.0
    trans = defaultdict(int)
    span = 0.0
    nk = 0
    for a, b in pairs:
        span += rows[b][1] - rows[a][1]
        nk += b - a
        seg = rows[a:b + 1]            # up to and including the next anchor: the gap in front of it belongs to this step
        cur_end, cur_name = seg[0][2], seg[0][0]
        for n, st, en in seg[1:]:
            trans[(short(cur_name), short(n))] += 1
            if st > cur_end:
                gaps[(short(cur_name), short(n))] += st - cur_end
                count[(short(cur_name), short(n))] += 1
                idle += st - cur_end
            if en > cur_end:
                cur_end, cur_name = en, n
    busy = span - idle
    print(f"{steps} steps: {span / steps / 1e3:.1f} us per step, busy {busy / steps / 1e3:.1f}, idle {idle / steps / 1e3:.1f} "
          f"({100 * idle / span:.1f} %), kernels per step {nk / steps:.1f}")
    print(f"{'before -> after':60s} {'us/step':>9s} {'gaps/step':>10s} {'us/gap':>8s} {'transitions/step':>17s}")
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1]):
        print(f"{k[0] + ' -> ' + k[1]:60s} {v / steps / 1e3:9.1f} {count[k] / steps:10.2f} {v / count[k] / 1e3:8.1f} {trans[k] / steps:17.2f}")


if __name__ == "__main__":
    main()
