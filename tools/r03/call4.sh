#!/bin/bash
# Parity and timing under the GSR_BLEND_BWD variants given as arguments.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "$@"; do
  echo "== parity with GSR_BLEND_BWD=$v"
  GSR_BLEND_BWD=$v timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_random_scenes.py -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider -k "backward or module_api or empty or random" -s > gpurun_out/r03_parity_$v.log 2>&1 || { tail -60 gpurun_out/r03_parity_$v.log; exit 1; }
  tail -2 gpurun_out/r03_parity_$v.log
done
for v in lanes "$@"; do
  echo "== timing with GSR_BLEND_BWD=$v" | tee -a gpurun_out/r03_blend_variants.log
  GSR_BLEND_BWD=$v timeout -k 10 300 python -u tools/exp_blend.py c3 c3w c5 --reps 20 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_blend_variants.log
done
