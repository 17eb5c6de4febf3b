#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: start offset,
gap to the previous kernel's end, duration and short name of every kernel between two
consecutive preprocess_bwd launches, plus span / busy time of the last few steps."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import short  # noqa: E402


def main(path, which=-2):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "preprocess_bwd" in r["Kernel_Name"]]
    for k in range(max(0, len(idx) - 6), len(idx) - 1):
        seg = rows[idx[k] + 1: idx[k + 1] + 1]
        t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
        print(f"step span {(t1 - t0) / 1e3:8.1f} us   busy {busy / 1e3:8.1f} us   kernels {len(seg)}")
    a, b = idx[which] + 1, idx[which + 1] + 1
    t0 = prev = int(rows[a]["Start_Timestamp"])
    print(f"{'start_us':>9s} {'gap_us':>7s} {'dur_us':>8s}  kernel")
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} {(s - prev) / 1e3:7.1f} {(e - s) / 1e3:8.1f}  {short(r['Kernel_Name'])}"
              f"  [grid {r['Grid_Size_X']}, vgpr {r['VGPR_Count']}, lds {r['LDS_Block_Size']}]")
        prev = e


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -2)
