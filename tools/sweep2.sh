#!/bin/bash
# diagnostics: bench cfg3 for "LANES:WG_PER_CU:TOKEN" combinations
for v in "$@"; do
  IFS=: read lanes wg tok <<< "$v"
  DIMN_LANES=$lanes DIMN_WG_PER_CU=$wg DIMN_TOKEN=$tok python bench.py --config cfg3 --epochs 2 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1])
print('lanes=$lanes wg_per_cu=$wg token=$tok  lane_step_ms=%.4f  w1_ms=%.4f w1_GBs=%.0f  impute_ms(2 epochs)=%.1f val=%.4f' % (r['config']['lane_step_ms'], r['roofline']['avg_launch_ms'], r['roofline']['achieved'], r['ms_per_step'], r['config']['final_val_loss']))"
done
