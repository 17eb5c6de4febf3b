# The whole GPU test suite on the box, log under gpurun_out/.   bash tools/gpu_tests.sh [TAG] [pytest args]
set -e
TAG=${1:-tests}; shift || true
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider --durations=15 "$@" > gpurun_out/${TAG}.log 2>&1 || { tail -40 gpurun_out/${TAG}.log; exit 1; }
tail -25 gpurun_out/${TAG}.log
