#!/bin/bash
# resident kernel check: parity tests of the R variants, then K=5 bench (new vs variants under tools/*.so) and the phase timeline
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03res; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_bf16.py -m gpu -q -x -k "R or resident or cfg4 or cfg2 or drift or path_choice or bf16_x_arena" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --epochs 6"
run() { # name lib args
  if [ -n "$2" ]; then export DIMN_LIB_PATH=$2; else unset DIMN_LIB_PATH; fi
  timeout 300 $B $3 > $O/$1.json 2> $O/$1.err
  unset DIMN_LIB_PATH
  python - <<PY
import json
try:
    d=json.load(open("$O/$1.json")); print("$1", round(d["value"]), "step us %.2f" % (1e3*d["config"]["lane_step_ms"]), "frac %.3f" % d["roofline"].get("frac"), d["config"]["final_val_loss"])
except Exception as e: print("$1 failed", e)
PY
}
for rep in 1 2; do
run k5_new_$rep "" "--limit-subnets 5"
for v in $VARIANTS; do run k5_${v}_$rep tools/libdimn_$v.so "--limit-subnets 5"; done
done
run k10_new "" "--limit-subnets 10"
run cfg2_new "" "--config cfg2"
timeout 600 python tools/res_timeline.py 5 > $O/timeline.txt 2>&1; tail -45 $O/timeline.txt
