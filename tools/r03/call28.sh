#!/bin/bash
# Wave priority inside a turn of the blend backward (s_setprio 2 ... 0): variant 1 raises it from the reduction's
# LDS stores to the atomic, variant 2 from the splat's LDS reads to the vote.  Built beforehand
# (-DGSR_EXPERIMENT_PRIO=1|2 on a blend.hip with the four s_setprio calls).
# Two alternations: variant 1 C3 231.8 -> 225.3 us, c3w 959 -> 923 us (the same at levels 1 and 3, and raised only
# behind the reduction's LDS reads; raised from the vote on 227 / 928); variant 2 no change.  Variant 1 is in.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=gaussian-splatting-own_amd
V=$GRAFT_REPO_ROOT/tools/experiments/variants
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload $1 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels']['blend_bwd']['ms'], d['train_step_ms'])"; }
for i in 1 2; do for wl in c3 c3w; do
echo "== shipped $wl"; timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
for v in ${VARIANTS:-prio1 prio2}; do
echo "== $v $wl"; GSR_HIP_LIB=$V/libgsr_$v.so timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
done; done; done
