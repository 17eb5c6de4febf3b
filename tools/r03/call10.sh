#!/bin/bash
# A/B of the depth segments on one box: blend kernels on c5r and c3, alternating.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
echo "== segments" && timeout -k 10 280 python3 -u tools/exp_blend.py c3 c5r --reps 40 2>&1 | grep -v amdgpu.ids
echo "== whole walks" && GSR_NO_SEGMENTS=1 timeout -k 10 280 python3 -u tools/exp_blend.py c3 c5r --reps 40 2>&1 | grep -v amdgpu.ids
done
echo "== step, segments" && timeout -k 10 280 python3 -u bench.py --no-cpu-baseline --no-variants --steps 200 --warmup 20 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['render_ms'], d['kernels']['blend_bwd']['ms'])"
echo "== step, whole walks" && GSR_NO_SEGMENTS=1 timeout -k 10 280 python3 -u bench.py --no-cpu-baseline --no-variants --steps 200 --warmup 20 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['render_ms'], d['kernels']['blend_bwd']['ms'])"
echo "== step, segments" && timeout -k 10 280 python3 -u bench.py --no-cpu-baseline --no-variants --steps 200 --warmup 20 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['render_ms'], d['kernels']['blend_bwd']['ms'])"
echo "== step, whole walks" && GSR_NO_SEGMENTS=1 timeout -k 10 280 python3 -u bench.py --no-cpu-baseline --no-variants --steps 200 --warmup 20 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['render_ms'], d['kernels']['blend_bwd']['ms'])"
