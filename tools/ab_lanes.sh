#!/bin/bash
# A/B on one box: sub-net lanes (DIMN_LANES) for the 4- and 2-GPU shares (K = 10 / 20 sub-nets per rank)
cd "$(dirname "$0")/.." || exit 1
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s cells/s %.0f  ms %.1f  step_us %.2f  val %.6f' % ('$1', d['value'], d['ms_per_step'], 1e3*d['config']['lane_step_ms'], d['config']['final_val_loss']))"; }
for k in 10 20; do for l in 1 2 3 4; do
  DIMN_LANES=$l python bench.py --limit-subnets $k --epochs 6 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | show "K=$k lanes=$l"
done; done
