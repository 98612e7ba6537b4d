#!/bin/bash
# round 3, GPU call 1: the whole -m gpu suite, then same-box baselines for this round's kernel work
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --epochs 6"
timeout 300 $B --limit-subnets 5 > $O/k5_coop.json 2> $O/k5_coop.err
DIMN_RES_COOP=0 timeout 300 $B --limit-subnets 5 > $O/k5_plain.json 2> $O/k5_plain.err
timeout 300 $B > $O/cfg3_e6.json 2> $O/cfg3_e6.err
timeout 300 $B --hidden 300 > $O/cfg3_h300_e6.json 2> $O/cfg3_h300.err
tail -5 $O/pytest.log
for f in k5_coop k5_plain cfg3_e6 cfg3_h300_e6; do python - <<PY
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"]), d["config"]["lane_step_ms"], d["roofline"].get("frac"))
except Exception as e: print("$f failed", e)
PY
done
