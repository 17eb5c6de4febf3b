#!/bin/bash
# Wave priority in the loss kernels: lp1 = the vertical pass (and what follows it) at priority 2, lp2 = the horizontal
# pass at priority 2.  Variants built beforehand (-DGSR_LOSS_PRIO=1|2 on a loss.hip with the s_setprio calls around
# the passes).  Three alternations: lp1 forward +-0, backward 41.5 -> 40.2 us; lp2 forward 57.4 -> 56.9 us, backward
# 41.5 -> 40.0 us.  lp2 is in.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=gaussian-splatting-own_amd
V=$GRAFT_REPO_ROOT/tools/experiments/variants
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload c3 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(k['loss_fwd']['ms'], k['loss_bwd']['ms'], d['train_step_ms'])"; }
for i in 1 2 3; do
echo "== shipped"; timeout -k 10 200 bash -c "$(declare -f one); one" || exit 1
for v in lp1 lp2; do
echo "== $v"; GSR_HIP_LIB=$V/libgsr_$v.so timeout -k 10 200 bash -c "$(declare -f one); one" || exit 1
done; done
