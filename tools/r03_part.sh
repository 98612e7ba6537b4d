#!/bin/bash
# CU-partitioned two-stream step (DIMN_PART): sweep of groups x CUs for the second layer, cfg3, 4 epochs
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03part; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --epochs 4"
run() { # name env
  env $2 timeout 300 $B > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$1.json")); print("$1", round(d["value"]), "step us %.2f" % (1e3*d["config"]["lane_step_ms"]), "B1F1 us %.1f frac %.3f" % (1e3*d["roofline"]["avg_launch_ms"], d["roofline"].get("frac")), d["config"]["final_val_loss"])
except Exception as e: print("$1 failed", e, open("$O/$1.err").read()[-300:])
PY
}
run base "X=1"
for spec in $SPECS; do
  g=${spec%%:*}; cm=${spec##*:}
  run part_g${g}_cm${cm} "DIMN_PART=$g DIMN_PART_CM=$cm"
done
run base2 "X=1"
