#!/bin/bash
# per-rank view of an N-GPU job (K/N sub-nets on this GPU): fused vs two-kernel second layer
for k in 5 10 20; do
  for mid in 1 0; do
    DIMN_MID=$mid python bench.py --no-cpu-baseline --limit-subnets $k --epochs 6 --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('K=$k DIMN_MID=$mid  step_ms %.4f  w1_launch_ms %.4f' % (d['config']['lane_step_ms'], r['avg_launch_ms']))"
  done
done
