#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03bf; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -m gpu -q -x -s -k "bf16 or cfg5 or path_choice" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -15
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --epochs 4"
run() { # name env args
  env $2 timeout 600 $B $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$1.json")); print("$1", round(d["value"]), "step us %.2f" % (1e3*d["config"]["lane_step_ms"]), "frac %.3f" % d["roofline"].get("frac"), d["config"]["final_val_loss"], d["roofline"]["kernel"][:30])
except Exception as e: print("$1 failed", e)
PY
}
run k5_bf16 "X=1" "--limit-subnets 5 --precision bf16"
run k5_bf16_fp32gemm "DIMN_TRAIN_BF16=0" "--limit-subnets 5 --precision bf16"
run cfg5_8_res "X=1" "--config cfg5 --cells 100000 --limit-subnets 8 --precision bf16 --stream"
run cfg5_8_stream "DIMN_RESIDENT=0" "--config cfg5 --cells 100000 --limit-subnets 8 --precision bf16 --stream"
