#!/bin/bash
# After a change to the blend kernels' arithmetic: the whole GPU suite, then the kernels' times.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread -p no:cacheprovider > gpurun_out/r03_suite4.log 2>&1 || { grep -n "FAILED\|Error\|assert " gpurun_out/r03_suite4.log | head -30; tail -30 gpurun_out/r03_suite4.log; exit 1; }
tail -2 gpurun_out/r03_suite4.log
for i in 1 2; do timeout -k 10 280 python3 -u tools/exp_blend.py c3 c3w c5 --reps 40 2>&1 | grep -v amdgpu.ids; done
