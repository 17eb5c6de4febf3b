#!/bin/bash
# First GPU call: the whole GPU suite on the default kernels, the backward parity tests and the blend
# timings under each formulation of the blend backward (GSR_BLEND_BWD).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== suite (default kernels)"; 
timeout -k 10 700 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -p no:cacheprovider --durations=12 -s > gpurun_out/r03_suite1.log 2>&1 || { tail -60 gpurun_out/r03_suite1.log; exit 1; }
tail -20 gpurun_out/r03_suite1.log
for v in rows8 rows4 rows8acc rows4acc; do
  echo "== parity with GSR_BLEND_BWD=$v"
  GSR_BLEND_BWD=$v timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider -k "backward or module_api or empty" -s > gpurun_out/r03_parity_$v.log 2>&1 || { tail -60 gpurun_out/r03_parity_$v.log; exit 1; }
  tail -3 gpurun_out/r03_parity_$v.log
done
for v in lanes rows8 rows4 rows8acc rows4acc; do
  echo "== timing with GSR_BLEND_BWD=$v"
  GSR_BLEND_BWD=$v timeout -k 10 200 python -u tools/exp_blend.py c3 c3w --reps 20 2>&1 | tee -a gpurun_out/r03_blend_variants.log
done
