#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_segments.py tests/test_gpu_robustness.py -x -q --timeout 120 --timeout-method thread > gpurun_out/seg4.log 2>&1 || { tail -40 gpurun_out/seg4.log; exit 1; }
tail -3 gpurun_out/seg4.log
one() { python3 -u bench.py --no-cpu-baseline --no-variants --steps 200 --warmup 20 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['render_ms'], d['kernels']['blend_bwd']['ms'], d['kernels']['blend_fwd']['ms'])"; }
for i in 1 2; do
echo "== step, segments" && timeout -k 10 280 bash -c "$(declare -f one); one"
echo "== step, whole walks" && GSR_NO_SEGMENTS=1 timeout -k 10 280 bash -c "$(declare -f one); one"
done
