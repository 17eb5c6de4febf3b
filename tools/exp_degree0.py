#!/usr/bin/env python3
"""The C3 scene with SH degree 0 (one coefficient per Gaussian): what the preprocess kernels take without the
192 bytes of SH coefficients per Gaussian.  Run under rocprofv3 --kernel-trace and read with tools/step_trace.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.WORKLOADS["c3d0"] = (1_000_000, 1920, 1080, 0, 0.0, None)
sys.argv = [sys.argv[0], "c3d0", "30"]
exec(open(os.path.join(ROOT, "tools", "exp_steps.py")).read())
