#!/bin/bash
# Wave priority elsewhere in the blend kernels: f2 = the forward's walk at priority 2 between a splat's LDS reads and
# the end of its turn; s3 / s1 = the staging of a batch (list entries, records, quadrant test) at priority 3 / 1 in
# both kernels.  The shipped library has the backward's reduction at priority 2.  Variants built beforehand.
# Two alternations: f2 forward C3 +-0, c3w -1.3 %, C5 +4 %; s3 / s1 backward C5 -2 %, C3 and c3w +-0, forward +1 %:
# none kept.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=gaussian-splatting-own_amd
V=$GRAFT_REPO_ROOT/tools/experiments/variants
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload $1 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(k['blend_fwd']['ms'], k['blend_bwd']['ms'], d['render_ms'], d['train_step_ms'])"; }
for i in 1 2; do for wl in c3 c3w c5; do
echo "== shipped $wl"; timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
for v in ${VARIANTS:-f2 s3 s1}; do
echo "== $v $wl"; GSR_HIP_LIB=$V/libgsr_$v.so timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
done; done; done
