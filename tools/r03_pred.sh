#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03pred; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_edges.py tests/test_gpu_configs.py -m gpu -q -x -k "bf16 or cfg5 or degenerate" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -8
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --epochs 2 --precision bf16"
run() { # name env
  env $2 timeout 300 $B > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$1.json")); p=d["roofline"]["predict"]; print("$1", round(d["value"]), "predict ms %.2f  TFLOP/s %.0f  frac %.3f" % (p["ms"], p["achieved"], p["frac"]), d["config"]["final_val_loss"])
except Exception as e: print("$1 failed", e, open("$O/$1.err").read()[-300:])
PY
}
run new "X=1"
run r2 "DIMN_PREDICT_R2=1"
for v in $VARIANTS; do run $v "DIMN_LIB_PATH=tools/libdimn_$v.so"; done
