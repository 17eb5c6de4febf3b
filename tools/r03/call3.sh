#!/bin/bash
# Backward parity and blend timings of the tree's library (and of other builds given as arguments).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_random_scenes.py -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider -k "backward or module_api or empty or random" -s > gpurun_out/r03_parity_tree.log 2>&1 || { tail -60 gpurun_out/r03_parity_tree.log; exit 1; }
tail -2 gpurun_out/r03_parity_tree.log
for lib in "" "$@"; do [ -n "$lib" ] && lib=$GRAFT_REPO_ROOT/$lib
  echo "== timing ${lib:-tree}" | tee -a gpurun_out/r03_blend_timing.log
  GSR_HIP_LIB=$lib timeout -k 10 300 python -u tools/exp_blend.py c3 c3w c5 --reps 20 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_blend_timing.log
done
