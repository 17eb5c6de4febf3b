#!/bin/bash
# Where a (wave, splat) turn of the blend backward spends its time, by ablation (variant libraries whose gradients
# are wrong on purpose, built beforehand in the container from blend.hip):
#   -DGSR_EXPERIMENT_NO_ATOMICS   everything but the nine atomic additions of a turn
#   -DGSR_EXPERIMENT_NO_REDUCE    neither the reduction through LDS nor the atomics: staging, culling and the per-pixel arithmetic
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/tools/experiments/variants
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload $1 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels']['blend_bwd']['ms'])"; }
for i in 1 2; do for wl in c3 c3w; do
echo "== shipped $wl"; timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
for v in noatomics noreduce; do
echo "== $v $wl"; GSR_HIP_LIB=$V/libgsr_$v.so timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
done; done; done
