#!/bin/bash
# The blend backward without its per-turn vote (98 % of the turns pass it, tools/r03/turn_probe.py): every turn
# runs to the end, a turn without a hit adds zeros.  Variant built beforehand: DEFS=-DGSR_EXPERIMENT_NO_VOTE
# Three alternations: C3 236.3 -> 237.1 us, c3w 935 -> 929 us, C5 541.6 -> 552.2 us: the vote stays.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=gaussian-splatting-own_amd
V=$GRAFT_REPO_ROOT/tools/experiments/variants
GSR_HIP_LIB=$V/libgsr_novote.so timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_segments.py tests/test_gpu_blend_formulations.py -x -q --timeout 300 --timeout-method thread > gpurun_out/novote_tests.log 2>&1 || { tail -20 gpurun_out/novote_tests.log; exit 1; }
tail -2 gpurun_out/novote_tests.log
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload $1 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels']['blend_bwd']['ms'], d['train_step_ms'])"; }
for i in 1 2 3; do for wl in c3 c3w c5; do
echo "== shipped $wl"; timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
echo "== no vote $wl"; GSR_HIP_LIB=$V/libgsr_novote.so timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
done; done
