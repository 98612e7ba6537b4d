#!/bin/bash
# diagnostics: bench cfg3 for combinations "LANES:WG_PER_CU" given as arguments
for v in "$@"; do
  IFS=: read lanes wg <<< "$v"
  DIMN_LANES=$lanes DIMN_WG_PER_CU=$wg python bench.py --config cfg3 --epochs 1 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1])
print('lanes=$lanes wg_per_cu=$wg  step_ms=%.4f  w1_kernel_ms=%.4f  epoch_total_ms=%.1f val=%.4f' % (r['config']['lane_step_ms'], r['roofline']['avg_launch_ms'], r['ms_per_step'], r['config']['final_val_loss']))"
done
