#!/bin/bash
# A/B of two builds of the library on one box: bench.py with the tree's library and with the one given.
#   bash tools/exp_ab_lib.sh /path/to/other/libgsr_hip.so [workload ...]
set -e
OTHER=$1; shift
for wl in ${@:-c3}; do
  for round in 1 2; do
    for lib in "" "$OTHER"; do
      GSR_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --workload $wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', '${lib:-tree}'.split('/')[-1], d['value'], d['ms_per_step'], d['render_ms'], 'blend_bwd', d['roofline']['kernel_ms'], {k: v['ms'] for k, v in d['kernels'].items() if k in ('blend_fwd', 'loss_fwd', 'loss_bwd', 'preprocess_bwd')})"
    done
  done
done
