#!/bin/bash
# configs[4] share at the FULL 1M cells (8 sub-nets, bf16, streamed), 2 epochs, one warm-up impute (all allocations happen there):
# the resident kernel's two large-arena measures -- DIMN_RES_SPLIT (tile order of its loop: 1 = gradient tiles first, two tile-times
# of request lead) and DIMN_RES_EPOCH_ROWS (1 = the epoch's rows copied into visiting order before the launch); last: the library's choice
cd "$(dirname "$0")/.." || exit 1
for v in "0 0" "1 0" "0 1" "1 1" ""; do
  echo "=== DIMN_RES_SPLIT / DIMN_RES_EPOCH_ROWS = $v"
  if [ -n "$v" ]; then set -- $v; export DIMN_RES_SPLIT=$1 DIMN_RES_EPOCH_ROWS=$2; else unset DIMN_RES_SPLIT DIMN_RES_EPOCH_ROWS; fi
  DIMN_TRACE=1 timeout 900 python bench.py --config cfg5 --limit-subnets 8 --precision bf16 --stream --steps 1 --warmup 1 --epochs 2 --no-cpu-baseline --no-dropin 2> gpurun_out/tr.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"lane_step_us %.2f  val %.6f  impute_s %.2f\" % (1e3*d[\"config\"][\"lane_step_ms\"], d[\"config\"][\"final_val_loss\"], d[\"ms_per_step\"]/1e3))"
  grep "resident epoch" gpurun_out/tr.err | head -1
done
