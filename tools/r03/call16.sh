#!/bin/bash
# The parity tests under the library's switches: stage calls, short lists, no segments, the one-pixel-per-lane backward.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T="tests/test_gpu_parity.py tests/test_gpu_robustness.py tests/test_gpu_fused_activations.py tests/test_gpu_segments.py tests/test_gpu_training_loop.py tests/test_gpu_random_scenes.py"
for env in "GSR_STAGED=1" "GSR_SHORT_LISTS=1" "GSR_NO_SEGMENTS=1" "GSR_BLEND_BWD=lanes" "GSR_SEPARATE_FILLS=1"; do
  echo "== $env"
  env $env timeout -k 10 400 python -u -m pytest $T -q --timeout 200 --timeout-method thread -p no:cacheprovider 2>&1 | tail -4
done
