#!/bin/bash
# New tests (segments at the shipped length, c5w at full size against the oracle), then the benchmark line.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_fullsize.py::test_full_size_against_oracle -x -q --timeout 400 --timeout-method thread -s > gpurun_out/seg3.log 2>&1 || { tail -40 gpurun_out/seg3.log; exit 1; }
grep -E "passed|failed|deepest|c5w" gpurun_out/seg3.log | cut -c1-250 | tail -24
timeout -k 10 600 python -u bench.py "$@" > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { tail -30 gpurun_out/r03_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "render_ms", "lists", "reference_operator_api") if k in d}))
print(json.dumps({k: v["ms"] for k, v in d["kernels"].items()}))
for k, v in d.get("variants", {}).items():
    print(k, v["render_ms"], v["train_step_ms"], v["lists"], v["stage_ms"])
PY
