#!/usr/bin/env python3
"""Compact table of a rocprofv3 --kernel-trace --stats run: shortens the
template-heavy kernel names and prints calls, total, average and share."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|merge_sort_block_merge_impl|"
                  r"merge_sort_block_sort|lookback_scan_kernel|init_lookback_scan_state|radix_sort_block_sort|"
                  r"onesweep\w*|scan\w*)", name)
    if "rocprim" in name and m:
        tag = "partition" if "partition_config" in name else ""
        return f"rocprim::{m.group(1)}{' ' + tag if tag else ''}"
    name = re.sub(r"<.*", "<...>", name)
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main(path):
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    total = sum(v[1] for v in agg.values())
    print(f"{'kernel':92s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'share%':>7s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:92s} {c:6d} {t / 1e6:10.3f} {t / c / 1e3:10.2f} {100 * t / total:7.2f}")
    print(f"{'TOTAL':92s} {'':6s} {total / 1e6:10.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
