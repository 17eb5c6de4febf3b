#!/bin/bash
# Loss kernels at other occupancy targets: the step's stage table per variant library, alternating.
# The variants are built beforehand, in the container (f = 6, 7):
#   make -C gaussian-splatting-own_amd/csrc OUT=../../tools/experiments/variants/libgsr_lossfwd$f.so OBJ=_obj_lossfwd$f DEFS=-DGSR_LOSS_FWD_PER_CU=$f
set -e
cd $GRAFT_REPO_ROOT
one() { python3 -u bench.py --no-cpu-baseline --no-variants --steps 200 --warmup 20 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k: d['kernels'][k]['ms'] for k in ('loss_fwd','loss_bwd')})"; }
for i in 1 2; do
for v in "" lossfwd6 lossfwd7; do
  echo "== ${v:-shipped}"
  if [ -z "$v" ]; then timeout -k 10 200 bash -c "$(declare -f one); one"; else GSR_HIP_LIB=$GRAFT_REPO_ROOT/tools/experiments/variants/libgsr_$v.so timeout -k 10 200 bash -c "$(declare -f one); one"; fi
done; done
