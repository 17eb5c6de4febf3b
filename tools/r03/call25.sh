#!/bin/bash
# The fused Adam step with one, two or four 16-byte accesses per array and thread, with and without the
# nontemporal hint (variants built beforehand: GSR_ADAM_ROUNDS, GSR_ADAM_NT); checksums must agree.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=gaussian-splatting-own_amd
V=$GRAFT_REPO_ROOT/tools/experiments/variants
for i in 1 2; do
echo "== shipped"; timeout -k 10 120 python3 -u tools/exp_adam.py 1000000 || exit 1
for v in r1 r2 r4 r1nt r2nt; do
echo "== $v"; GSR_HIP_LIB=$V/libgsr_adam_$v.so timeout -k 10 120 python3 -u tools/exp_adam.py 1000000 || exit 1
done; done
echo "== 6M"; timeout -k 10 120 python3 -u tools/exp_adam.py 6000000
for v in r2 r2nt; do echo "== 6M $v"; GSR_HIP_LIB=$V/libgsr_adam_$v.so timeout -k 10 120 python3 -u tools/exp_adam.py 6000000 || exit 1; done
