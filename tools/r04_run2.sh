#!/bin/bash
# round 4, GPU call 2: the whole -m gpu suite; the resident kernel after the re-arm reordering (K = 5, 10); the general path with
# Adam fused behind its gradient kernels; hidden 300 for reference
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2700 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1
echo "pytest rc=$?"; tail -8 $O/tests.log
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy"
run() { # name args
  timeout 900 $B $2 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$1.json")); c=d["config"]; print("%-22s %8.0f cells/s  step wall %.4f ms  lane_step %.4f ms  frac %.3f  val %.4f" % ("$1", d["value"], c["train_step_ms_wall"], c["lane_step_ms"], d["roofline"].get("frac") or 0, c["final_val_loss"]))
except Exception as e: print("$1 failed", e, open("$O/$1.err").read()[-400:])
PY
}
for rep in 1 2; do run k5_res_$rep "--limit-subnets 5 --epochs 6"; done
run k10_res "--limit-subnets 10 --epochs 6"
run k5_bf16 "--limit-subnets 5 --epochs 6 --precision bf16"
run h256 "--epochs 4"
run h256_general "--general --epochs 4"
run b128_general "--batch 128 --epochs 4"
run h512_b128_general "--batch 128 --hidden 512 --epochs 4"
run h300 "--hidden 300 --epochs 4"
