#!/bin/bash
# resident kernel, bf16 operands: tile order of the loop (DIMN_RES_SPLIT=1: gradient tiles first, two tile-times of request lead;
# 0: alternating) at K = 5 of configs[3] and at configs[4]'s 8 sub-nets with 200k cells; one box
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2; do for v in 1 0; do
  echo "=== DIMN_RES_SPLIT=$v rep $rep"
  for a in "--limit-subnets 5 --epochs 6" "--config cfg5 --cells 200000 --limit-subnets 8 --stream --epochs 2"; do
    DIMN_RES_SPLIT=$v timeout 300 python bench.py $a --precision bf16 --no-cpu-baseline --steps 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"lane_step_us %.2f  val %.6f\" % (1e3*d[\"config\"][\"lane_step_ms\"], d[\"config\"][\"final_val_loss\"]))"
  done
done; done
