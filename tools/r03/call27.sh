#!/bin/bash
# Nontemporal accesses in the preprocess: its input loads, its small output stores, its record stores, all three
# (variants built beforehand: GSR_PRE_NT_LOADS, GSR_PRE_NT_STORES, GSR_PRE_NT_RECORDS).
cd $GRAFT_REPO_ROOT
export PYTHONPATH=gaussian-splatting-own_amd
V=$GRAFT_REPO_ROOT/tools/experiments/variants
one() { python3 -u bench.py --no-cpu-baseline --no-variants --workload $1 --steps 100 --warmup 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(' '.join(f'{n} {k[n][\"ms\"]:.4f}' for n in ('preprocess_fwd','bin_sort','bin_lists','blend_fwd','loss_fwd','loss_bwd','blend_bwd','preprocess_bwd')), 'render', d['render_ms'], 'step', d['train_step_ms'])"; }
for i in 1 2; do for wl in c3 c5; do
echo "== shipped $wl"; timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
for v in ${VARIANTS:-pre_ntl pre_nts pre_ntr pre_ntlsr}; do
echo "== $v $wl"; GSR_HIP_LIB=$V/libgsr_$v.so timeout -k 10 200 bash -c "$(declare -f one); one $wl" || exit 1
done; done; done
