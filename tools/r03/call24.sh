#!/bin/bash
# The preprocess with its SH coefficients requested right after the near-plane test instead of after the projection,
# covariance and radius (the loads sit under a divergent branch, so the compiler cannot move them up itself).
# Variant built beforehand: DEFS=-DGSR_EXPERIMENT_EARLY_SH (sh[] declared at the top of the kernel, load_sh called
# behind "vis = pv.z > GSR_NEAR_CULL"; 90 registers instead of 87, same occupancy, results bit-identical).
# Three alternations: C3 77.9 -> 87.5 us, C5 370.6 -> 395.1 us.  Slower: rejected.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=gaussian-splatting-own_amd
V=$GRAFT_REPO_ROOT/tools/experiments/variants
GSR_HIP_LIB=$V/libgsr_earlysh.so timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fused_activations.py -x -q --timeout 300 --timeout-method thread > gpurun_out/earlysh_tests.log 2>&1 || { tail -20 gpurun_out/earlysh_tests.log; exit 1; }
tail -2 gpurun_out/earlysh_tests.log
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload $1 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels']['preprocess_fwd']['ms'], d['render_ms'], d['train_step_ms'])"; }
for i in 1 2 3; do for wl in c3 c5; do
echo "== shipped $wl"; timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
echo "== early SH $wl"; GSR_HIP_LIB=$V/libgsr_earlysh.so timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
done; done
