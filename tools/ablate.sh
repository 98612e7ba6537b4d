#!/bin/bash
# diagnostics: time k_w1_update_fwd ablation variants (DIMN_DBG bits) on cfg3
for d in 0 1 2 4 8 16 32 6 14 17 49 63; do
  DIMN_DBG=$d python bench.py --config cfg3 --epochs 1 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1])
print('DBG=%-3s w1_kernel_ms=%.4f step_ms=%.4f' % ('$d', r['roofline']['avg_launch_ms'], r['config']['lane_step_ms']))"
done
