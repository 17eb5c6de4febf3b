# kernel statistics of the two other single-GPU workloads (C5: 6M Gaussians with hot tiles, C2: 100k at 800x800)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload c5 > gpurun_out/prof_c5.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_c5/*/*_kernel_stats.csv > gpurun_out/c5_stats.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload c2 > gpurun_out/prof_c2.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_c2/*/*_kernel_stats.csv > gpurun_out/c2_stats.txt
tail -1 gpurun_out/prof_c5.log | cut -c1-200
tail -1 gpurun_out/prof_c2.log | cut -c1-200
