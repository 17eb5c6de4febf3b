#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc run (csv): one line per kernel with the
average counter value per dispatch.  Only kernels matching the optional
substring filters are shown."""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import short  # noqa: E402


def main(directory, *filters):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if filters and not any(s in k for s in filters):
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            s, n = acc[k][c]
            print(f"    {c:32s} mean/dispatch {s / n:18.1f}   dispatches {n}")


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
