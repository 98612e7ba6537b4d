#!/bin/bash
# diagnostics: time B1F1 variants (DIMN_B1F1) x work-table density (DIMN_WG_PER_CU) on cfg3
for v in "$@"; do
  IFS=: read var wg <<< "$v"
  DIMN_B1F1=$var DIMN_WG_PER_CU=$wg python bench.py --config cfg3 --epochs 1 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1])
print('variant=$var wg_per_cu=$wg  w1_kernel_ms=%.4f step_ms=%.4f val=%.4f' % (r['roofline']['avg_launch_ms'], r['config']['lane_step_ms'], r['config']['final_val_loss']))"
done
