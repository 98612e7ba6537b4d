#!/bin/bash
# does the first process on a fresh box run slower (clock / power ramp)?  same command three times, then warmup 4
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1  cells/s %.0f  step_ms %.4f  w1_launch_ms %.4f  frac %.3f' % (d['value'], d['config']['lane_step_ms'], r['avg_launch_ms'], r['frac']))"; }
for i in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | show "run$i warmup1"; done
python bench.py --no-cpu-baseline --warmup 4 2>/dev/null | show "run4 warmup4"
