#!/bin/bash
# A/B of an environment switch on one box: tools/ab_env.sh VAR "v1 v2 ..." [reps]
VAR=$1; VALS=$2; REPS=${3:-2}
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1  cells/s %.0f  step_ms %.4f  w1_launch_ms %.4f  frac %.3f  val %.6f' % (d['value'], d['config']['lane_step_ms'], r['avg_launch_ms'], r['frac'], d['config']['final_val_loss']))"; }
for rep in $(seq $REPS); do for v in $VALS; do env $VAR=$v python bench.py --no-cpu-baseline 2>/dev/null | show "$VAR=$v"; done; done
