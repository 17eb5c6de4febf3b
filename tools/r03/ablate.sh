#!/bin/bash
# Blend timings of the tree's library and of variant libraries (paths relative to the repository root); no parity run:
# ablation builds give wrong results on purpose.
cd $GRAFT_REPO_ROOT
for lib in "" "$@"; do [ -n "$lib" ] && lib=$GRAFT_REPO_ROOT/$lib
  echo "== timing ${lib:-tree}"
  GSR_HIP_LIB=$lib timeout -k 10 300 python -u tools/exp_blend.py c3 c3w --reps 20 2>&1 | grep -v amdgpu.ids
done
