# exp_blend.py once per library variant under tools/experiments/ (and the built library itself).  bash tools/exp_variants.sh "c3 c5" base v1 ...
set -e
cd $GRAFT_REPO_ROOT
WL=$1; shift
echo "== built"; timeout -k 10 200 python3 tools/exp_blend.py $WL
for v in "$@"; do echo "== $v"; GSR_HIP_LIB=$PWD/tools/experiments/libgsr_$v.so timeout -k 10 200 python3 tools/exp_blend.py $WL; done
