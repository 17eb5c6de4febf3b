# kernel times of the depth sort alone from a kernel trace.   bash tools/exp_sort_trace.sh TAG [lib]
set -e
TAG=$1; LIB=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$LIB" ]; then export GSR_HIP_LIB=$PWD/tools/experiments/libgsr_$LIB.so; fi
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 tools/exp_sort.py c3 > gpurun_out/prof_$TAG.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_$TAG/*/*_kernel_stats.csv | grep -E "sort_|kernel  " 
