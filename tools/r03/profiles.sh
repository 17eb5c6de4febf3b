#!/bin/bash
# The profile set: kernel trace + counter passes on C3 and c3w, kernel traces of C5, C2, c5w, c5r and of the step through
# the reference's operator API, then the default benchmark line.
set -e
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03 c3
bash tools/profile_round.sh r03_c3w c3w
bash tools/profile_workloads.sh r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_c5w -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --workload c5w > gpurun_out/prof_r03_c5w.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_r03_c5w/*/*_kernel_stats.csv > gpurun_out/r03_c5w_stats.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_c5r -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --workload c5r > gpurun_out/prof_r03_c5r.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_r03_c5r/*/*_kernel_stats.csv > gpurun_out/r03_c5r_stats.txt
bash tools/profile_unfused.sh
timeout -k 10 600 python3 bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || { tail -20 gpurun_out/r03_bench_default.err; exit 1; }
for wl in c2 c5; do timeout -k 10 300 python3 bench.py --workload $wl --no-cpu-baseline > gpurun_out/r03_bench_$wl.json 2>/dev/null; done
tail -c 600 gpurun_out/r03_bench_default.json
