#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (csv output): per kernel, calls and mean/total of each counter.

usage: tools/pmc_summary.py <dir-with-*counter_collection.csv> [out.txt]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB?  No: gfx950 derives them as
TCC_EA0_RDREQ-based *kilobytes* in some ROCm builds and bytes in others, so the raw number is printed and
bench.py applies the calibration (MI355X_MICROARCH.md, HBM section: FETCH_SIZE under-reports wide
coalesced reads by 2x on gfx950).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter_collection.csv under", d)
        return
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            a = acc[k][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    lines = [f"{'kernel':70s} {'counter':>12s} {'calls':>7s} {'mean':>16s} {'total':>18s}"]
    for k in sorted(acc, key=lambda k: -max(v[1] for v in acc[k].values())):
        for c, (n, s) in acc[k].items():
            kk = k if len(k) <= 70 else k[:67] + "..."
            lines.append(f"{kk:70s} {c:>12s} {n:7d} {s/n:16.1f} {s:18.1f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    if len(sys.argv) > 3:   # machine-readable: {kernel: {counter: {"calls": n, "mean": m}}}, merged into an existing file
        import json
        js = {}
        if os.path.exists(sys.argv[3]):
            js = json.load(open(sys.argv[3]))
        for k in acc:
            for c, (n, s) in acc[k].items():
                js.setdefault(k, {})[c] = {"calls": n, "mean": s / n}
        json.dump(js, open(sys.argv[3], "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
