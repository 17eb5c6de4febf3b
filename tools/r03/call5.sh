#!/bin/bash
# Whole parity file + random scenes + timing, with the blend kernels chosen by "FWD:BWD" pairs given as arguments.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "$@"; do
  f=${v%%:*}; b=${v##*:}
  echo "== parity with fwd=$f bwd=$b"
  GSR_BLEND_FWD=$f GSR_BLEND_BWD=$b timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_random_scenes.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 200 --timeout-method thread -p no:cacheprovider -s > gpurun_out/r03_parity_${f}_${b}.log 2>&1 || { tail -60 gpurun_out/r03_parity_${f}_${b}.log; exit 1; }
  tail -2 gpurun_out/r03_parity_${f}_${b}.log
done
for v in "lanes:lanes" "$@"; do
  f=${v%%:*}; b=${v##*:}
  echo "== timing with fwd=$f bwd=$b" | tee -a gpurun_out/r03_blend_variants.log
  GSR_BLEND_FWD=$f GSR_BLEND_BWD=$b timeout -k 10 300 python -u tools/exp_blend.py c2 c3 c3w c5 --reps 20 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_blend_variants.log
done
