#!/bin/bash
# The loss kernels with the waves' shares of the horizontal pass alternating from tile to tile (the 336 tasks of a
# tile are a round of 256 and a round of 80: without it the same two waves, hence the same two SIMDs, always take
# the second round).  Variant built beforehand: make ... DEFS=-DGSR_LOSS_TURN=128
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/tools/experiments/variants
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload c3 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(k['loss_fwd']['ms'], k['loss_bwd']['ms'], d['train_step_ms'])"; }
for i in 1 2 3; do
echo "== shipped"; timeout -k 10 200 bash -c "$(declare -f one); one" || exit 1
echo "== turn"; GSR_HIP_LIB=$V/libgsr_lossturn.so timeout -k 10 200 bash -c "$(declare -f one); one" || exit 1
done
