# kernel statistics of the step through the reference's unchanged operator API (PyTorch activations feeding GaussianRasterizer)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_unfused -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --unfused > gpurun_out/prof_unfused.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_unfused/*/*_kernel_stats.csv > gpurun_out/unfused_stats.txt
tail -1 gpurun_out/prof_unfused.log | cut -c1-300
