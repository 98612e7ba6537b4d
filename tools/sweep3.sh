#!/bin/bash
# diagnostics: bench cfg3 (2 epochs) for "LANES:MF:MB" combinations; prints end-to-end ms
for v in "$@"; do
  IFS=: read lanes mf mb <<< "$v"
  DIMN_LANES=$lanes DIMN_MF=$mf DIMN_MB=$mb python bench.py --config cfg3 --epochs 2 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1])
print('lanes=$lanes mf=$mf mb=$mb  impute_ms(2 epochs)=%.1f  lane_step_ms=%.4f  w1_ms=%.4f w1_GBs=%.0f' % (r['ms_per_step'], r['config']['lane_step_ms'], r['roofline']['avg_launch_ms'], r['roofline']['achieved']))"
done
