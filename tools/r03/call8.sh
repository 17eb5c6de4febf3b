#!/bin/bash
# Depth segments, second take: tests, then the probe with and without.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_segments.py -x -q --timeout 120 --timeout-method thread -s > gpurun_out/seg2.log 2>&1 || { tail -40 gpurun_out/seg2.log; exit 1; }
grep -E "passed|failed|segments of|dumps" gpurun_out/seg2.log | cut -c1-300 | tail -12
echo "== probe, segments" && timeout -k 10 280 python3 -u tools/r03/skew_probe.py 0.1 0.02 0.005 2>&1 | tee gpurun_out/seg_probe_on.log | sed 's/lists {.*}  depth/depth/; s/lists {.*}  whole/whole/'
echo "== probe, short lists, segments" && GSR_SHORT_LISTS=1 timeout -k 10 280 python3 -u tools/r03/skew_probe.py 0.1 0.02 0.005 2>&1 | tee gpurun_out/seg_probe_short_on.log | sed 's/lists {.*}  depth/depth/; s/lists {.*}  whole/whole/'
echo "== blend kernels, segments" && timeout -k 10 280 python3 -u tools/exp_blend.py c3 c3w c5 c5w 2>&1 | tee gpurun_out/seg_blend_on.log
