# exp_loss.py for the built library and for variants under tools/experiments/.   bash tools/exp_loss_variants.sh v1 v2 ...
set -e
cd $GRAFT_REPO_ROOT
echo "== built"; timeout -k 10 200 python3 tools/exp_loss.py
for v in "$@"; do echo "== $v"; GSR_HIP_LIB=$PWD/tools/experiments/libgsr_$v.so timeout -k 10 200 python3 tools/exp_loss.py; done
