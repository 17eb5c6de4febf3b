#!/bin/bash
# The blend backward with a second copy of its loop for stretches of a tile's list that every pixel of the half
# tile blends (no comparison with the pixels' own ends there).  Variant built beforehand in the container.
# Three alternations on one box: C3 239.1 -> 237.1 us, c3w 954.2 -> 944.1 us (-1 %); kept.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=gaussian-splatting-own_amd
V=$GRAFT_REPO_ROOT/tools/experiments/variants
GSR_HIP_LIB=$V/libgsr_twoloops.so timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_segments.py tests/test_gpu_blend_formulations.py tests/test_gpu_random_scenes.py -x -q --timeout 300 --timeout-method thread > gpurun_out/twoloops_tests.log 2>&1 || { tail -20 gpurun_out/twoloops_tests.log; exit 1; }
tail -2 gpurun_out/twoloops_tests.log
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload $1 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels']['blend_bwd']['ms'], d['train_step_ms'])"; }
for i in 1 2 3; do for wl in c3 c3w; do
echo "== shipped $wl"; timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
echo "== two loops $wl"; GSR_HIP_LIB=$V/libgsr_twoloops.so timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
done; done
