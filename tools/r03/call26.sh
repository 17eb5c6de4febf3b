#!/bin/bash
# Nontemporal accesses: more rounds of the fused Adam step with them, and the zero-fills that ride on the blend
# kernels stored nontemporally (variants built beforehand).
cd $GRAFT_REPO_ROOT
export PYTHONPATH=gaussian-splatting-own_amd
V=$GRAFT_REPO_ROOT/tools/experiments/variants
for i in 1 2; do
echo "== adam shipped"; timeout -k 10 120 python3 -u tools/exp_adam.py 1000000 || exit 1
for v in r2nt r3nt r4nt; do
echo "== adam $v"; GSR_HIP_LIB=$V/libgsr_adam_$v.so timeout -k 10 120 python3 -u tools/exp_adam.py 1000000 || exit 1
done; done
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload $1 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(k['blend_fwd']['ms'], k['blend_bwd']['ms'], k['preprocess_bwd']['ms'], d['render_ms'], d['train_step_ms'])"; }
for i in 1 2 3; do for wl in c3 c5; do
echo "== shipped $wl"; timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
echo "== nontemporal fills $wl"; GSR_HIP_LIB=$V/libgsr_fillnt.so timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
done; done
