#!/bin/bash
# Kernel trace of the default step with the template arguments kept: one line per distinct kernel instance.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_names -- python3 bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > gpurun_out/prof_names.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_names/*/*_kernel_trace.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
for k, v in rows[:40]:
    print("%8.2f us x %4d  %s" % (sum(v) / len(v), len(v), k[:150]))
PY
