#!/bin/bash
# A wider sweep of the seeded random scenes (SEEDS=<n> seeds, 300 unless set, instead of the suite's 16), once with the shipped settings
# and once with depth segments of 256 entries forced, so that segment boundaries fall inside small odd-sized scenes.
set -e
export PYTHONPATH=gaussian-splatting-own_amd
mkdir -p gpurun_out
GSR_RANDOM_SCENES=${SEEDS:-300} timeout -k 10 900 python -u -m pytest tests/test_gpu_random_scenes.py -x -q --timeout 120 --timeout-method thread \
    > gpurun_out/sweep_default.log 2>&1 && tail -3 gpurun_out/sweep_default.log
GSR_RANDOM_SCENES=${SEEDS:-300} GSR_RANDOM_SEGMENTS=256 timeout -k 10 900 python -u -m pytest tests/test_gpu_random_scenes.py -x -q --timeout 120 \
    --timeout-method thread > gpurun_out/sweep_segments.log 2>&1 && tail -3 gpurun_out/sweep_segments.log
