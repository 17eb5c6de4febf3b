#!/bin/bash
# What the blend backward's marks of the Gaussians it stages cost it (a byte store per staged splat with a non-empty
# quadrant mask): 4 us of 235 at C3, 6 us of 945 on c3w; they save the per-Gaussian backward 25 us at C3.
# The variant is built beforehand, in the container, from blend.hip with those stores under
#   #ifndef GSR_EXPERIMENT_NO_TOUCHED:
#   make -C gaussian-splatting-own_amd/csrc OUT=../../tools/experiments/variants/libgsr_notouched.so OBJ=_obj_notouched DEFS=-DGSR_EXPERIMENT_NO_TOUCHED
cd $GRAFT_REPO_ROOT
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload $1 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels']['blend_bwd']['ms'])"; }
for i in 1 2; do for wl in c3 c3w; do
echo "== shipped $wl"; timeout -k 10 200 bash -c "$(declare -f one); one $wl"
echo "== no touched marks $wl"; GSR_HIP_LIB=$GRAFT_REPO_ROOT/tools/experiments/variants/libgsr_notouched.so timeout -k 10 200 bash -c "$(declare -f one); one $wl"
done; done
