#!/bin/bash
# the families the CLI reaches -- hidden = 300 (the reference CLI's default width) and the general path
# (--batch-size 128; hidden 512), cfg3 shapes, 4 epochs, same box; plus the tuned default for reference
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03fam; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --epochs 4"
run() { # name args
  timeout 900 $B $2 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$1.json")); c=d["config"]; print("%-22s %8.0f cells/s  step wall %.3f ms  lane_step %.3f ms  B1F1 %.3f ms frac %.3f  val %.4f" % ("$1", d["value"], c["train_step_ms_wall"], c["lane_step_ms"], d["roofline"].get("avg_launch_ms") or 0, d["roofline"].get("frac") or 0, c["final_val_loss"]))
except Exception as e: print("$1 failed", e, open("$O/$1.err").read()[-400:])
PY
}
run h256 ""
run h300 "--hidden 300"
run h256_general "--general"
run b128_general "--batch 128"
run h512_b128_general "--batch 128 --hidden 512"
