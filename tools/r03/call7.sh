#!/bin/bash
# Depth segments of the blend backward: the low-opacity probe and the benchmark's scenes, with and without.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== probe, segments" && timeout -k 10 280 python3 -u tools/r03/skew_probe.py 0.1 0.02 0.005 2>&1 | tee gpurun_out/seg_probe_on.log | cut -c1-900
echo "== probe, whole walks" && GSR_NO_SEGMENTS=1 timeout -k 10 280 python3 -u tools/r03/skew_probe.py 0.1 0.02 0.005 2>&1 | tee gpurun_out/seg_probe_off.log | cut -c1-900
echo "== probe, short lists, segments" && GSR_SHORT_LISTS=1 timeout -k 10 280 python3 -u tools/r03/skew_probe.py 0.02 0.005 2>&1 | tee gpurun_out/seg_probe_short_on.log | cut -c1-900
echo "== blend kernels, segments" && timeout -k 10 280 python3 -u tools/exp_blend.py c3 c3w c5 c5w 2>&1 | tee gpurun_out/seg_blend_on.log
echo "== blend kernels, whole walks" && GSR_NO_SEGMENTS=1 timeout -k 10 280 python3 -u tools/exp_blend.py c3 c3w c5 c5w 2>&1 | tee gpurun_out/seg_blend_off.log
