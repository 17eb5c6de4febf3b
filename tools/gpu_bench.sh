# bench.py on the box, output under gpurun_out/.   bash tools/gpu_bench.sh TAG [bench args]
set -e
TAG=${1:-bench}; shift || true
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 bench.py "$@" > gpurun_out/${TAG}.json 2> gpurun_out/${TAG}.err || { tail -30 gpurun_out/${TAG}.err; exit 1; }
cat gpurun_out/${TAG}.json
