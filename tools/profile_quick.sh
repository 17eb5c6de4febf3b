# kernel statistics of a short C3 run (no counters).   bash tools/profile_quick.sh TAG [bench args]
set -e
TAG=${1:-q}; shift || true
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants "$@" > gpurun_out/prof_$TAG.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_$TAG/*/*_kernel_stats.csv > gpurun_out/${TAG}_stats.txt
head -24 gpurun_out/${TAG}_stats.txt
