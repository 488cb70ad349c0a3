#!/usr/bin/env python3
"""Mean kernel durations of a rocprofv3 --kernel-trace --pmc pass (csv output), merged into the counters' JSON as the
pseudo-counter DURATION_NS: counters are normalised by the duration of the pass they were taken in (the clock under counter
collection is not the free-running one).

usage: tools/pmc_durations.py <dir-with-*kernel_trace.csv> <counters.json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel_trace.csv under", d)
        return
    acc = defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"]]
            a[0] += 1
            a[1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    js = json.load(open(out)) if os.path.exists(out) else {}
    tag = "DURATION_NS_" + os.path.basename(os.path.normpath(d))
    for k, (n, s) in acc.items():
        js.setdefault(k, {})[tag] = {"calls": n, "mean": s / n}
    json.dump(js, open(out, "w"), indent=0, sort_keys=True)
    print("durations of", len(acc), "kernels ->", out)


if __name__ == "__main__":
    main()
