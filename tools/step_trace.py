#!/usr/bin/env python3
"""One steady train step out of a rocprofv3 kernel trace: every kernel with its duration and the gap before it.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/exp_steps.py c3 60
    python3 tools/step_trace.py gpurun_out/trace [step]"""
import csv
import glob
import sys

root = sys.argv[1]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(glob.glob(root + "/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "preprocess_bwd" in r["Kernel_Name"]]
a, b = ends[step], ends[step + 1]
prev = int(rows[a]["End_Timestamp"])
t0, busy = prev, 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{(s - prev) / 1e3:7.2f} gap {(e - s) / 1e3:8.2f} us  {name[:70]}")
    busy += e - s
    prev = e
print(f"kernels {busy / 1e3:.1f} us, gaps {(prev - t0 - busy) / 1e3:.1f} us, step {(prev - t0) / 1e3:.1f} us")
