#!/bin/bash
# The whole GPU suite, then the benchmark line (short).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -p no:cacheprovider --durations=8 -s > gpurun_out/r03_suite.log 2>&1 || { grep -n "FAILED\|Error\|assert" gpurun_out/r03_suite.log | head -40; tail -30 gpurun_out/r03_suite.log; exit 1; }
tail -12 gpurun_out/r03_suite.log
timeout -k 10 600 python -u bench.py "$@" > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { tail -30 gpurun_out/r03_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "render_ms", "lists", "reference_operator_api") if k in d}))
print(json.dumps(d["roofline"])[:1500])
print(json.dumps({k: v["ms"] for k, v in d["kernels"].items()}))
for k, v in d.get("variants", {}).items():
    print(k, v["render_ms"], v["train_step_ms"], v["lists"], v["stage_ms"])
print(d.get("cpu_baseline"))
PY
