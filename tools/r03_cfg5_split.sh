#!/bin/bash
# configs[4] share at the FULL 1M cells (8 sub-nets, bf16, streamed), 2 epochs: tile order of the resident kernel's loop
# (DIMN_RES_SPLIT=0: alternating, one tile-time of request lead; 1: gradient tiles first, two tile-times; unset: the library's choice)
cd "$(dirname "$0")/.." || exit 1
for v in 0 1 ""; do
  echo "=== DIMN_RES_SPLIT=$v"
  if [ -n "$v" ]; then export DIMN_RES_SPLIT=$v; else unset DIMN_RES_SPLIT; fi
  timeout 900 python bench.py --config cfg5 --limit-subnets 8 --precision bf16 --stream --steps 1 --warmup 0 --epochs 2 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"lane_step_us %.2f  val %.6f  impute_s %.2f\" % (1e3*d[\"config\"][\"lane_step_ms\"], d[\"config\"][\"final_val_loss\"], d[\"ms_per_step\"]/1e3))"
done
