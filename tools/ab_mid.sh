#!/bin/bash
# A/B of the second-layer paths on one box: DIMN_MID=1 (fused MFB + RED2) vs DIMN_MID=0 (MF + MB)
for rep in 1 2; do
  for mid in 1 0; do
    DIMN_MID=$mid python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('DIMN_MID=$mid  cells/s %.0f  step_ms %.4f  w1_launch_ms %.4f  frac %.3f  val %.6f' % (d['value'], d['config']['lane_step_ms'], r['avg_launch_ms'], r['frac'], d['config']['final_val_loss']))"
  done
done
