# kernel statistics of the other single-GPU workloads (C5: 6M Gaussians with hot tiles, C2: 100k at 800x800)
#   bash tools/profile_workloads.sh TAG
set -e
TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for WL in c5 c2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$WL -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --workload $WL > gpurun_out/prof_${TAG}_$WL.log 2>&1
  python3 tools/summarize_rocprof.py gpurun_out/prof_${TAG}_$WL/*/*_kernel_stats.csv > gpurun_out/${TAG}_${WL}_stats.txt
  tail -1 gpurun_out/prof_${TAG}_$WL.log | cut -c1-200
done
