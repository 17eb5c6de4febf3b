#!/bin/bash
# Does the dump code in the forward blend cost anything when no dump is taken?  The shipped library against a build
# with that code compiled out (both without segments at run time), alternating, on one box.
# The variant is built beforehand, in the container:
#   make -C gaussian-splatting-own_amd/csrc OUT=../../tools/experiments/variants/libgsr_nodumps.so OBJ=_obj_nodumps DEFS=-DGSR_EXPERIMENT_NO_FWD_DUMPS
set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo "== shipped" && GSR_NO_SEGMENTS=1 timeout -k 10 200 python3 -u tools/exp_blend.py c3 c3w --reps 40 2>&1 | grep -v amdgpu.ids
echo "== dump code compiled out" && GSR_NO_SEGMENTS=1 GSR_HIP_LIB=$GRAFT_REPO_ROOT/tools/experiments/variants/libgsr_nodumps.so timeout -k 10 200 python3 -u tools/exp_blend.py c3 c3w --reps 40 2>&1 | grep -v amdgpu.ids
done
