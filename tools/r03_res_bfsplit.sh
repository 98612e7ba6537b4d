#!/bin/bash
# resident kernel, bf16 operands: gradient tiles first then forward tiles (two tile-times of request lead) vs alternating; one box
cd "$(dirname "$0")/.." || exit 1
bash tools/ab_lib.sh 'for a in "--limit-subnets 5 --epochs 6" "--config cfg5 --cells 200000 --limit-subnets 8 --stream --epochs 2"; do timeout 300 python bench.py $a --precision bf16 --no-cpu-baseline --steps 1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"lane_step_us %.2f  val %.6f\" % (1e3*d[\"config\"][\"lane_step_ms\"], d[\"config\"][\"final_val_loss\"]))"; done' "-DDIMN_RES_BF_SPLIT=1" "-DDIMN_RES_BF_SPLIT=0"
rm -f deepimpute_amd/csrc/libdimn_ab*.so
