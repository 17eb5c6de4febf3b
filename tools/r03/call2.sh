#!/bin/bash
# Parity of the backward under each formulation of the blend backward (GSR_BLEND_BWD), then their timings.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in rows8 rows4 rows8acc rows4acc; do
  echo "== parity with GSR_BLEND_BWD=$v"
  GSR_BLEND_BWD=$v timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_random_scenes.py -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider -k "backward or module_api or empty or random" -s > gpurun_out/r03_parity_$v.log 2>&1 || { tail -60 gpurun_out/r03_parity_$v.log; exit 1; }
  tail -3 gpurun_out/r03_parity_$v.log
done
rm -f gpurun_out/r03_blend_variants.log
for v in lanes rows8 rows4 rows8acc rows4acc; do
  echo "== timing with GSR_BLEND_BWD=$v" | tee -a gpurun_out/r03_blend_variants.log
  GSR_BLEND_BWD=$v timeout -k 10 200 python -u tools/exp_blend.py c3 c3w --reps 20 2>&1 | tee -a gpurun_out/r03_blend_variants.log
done
