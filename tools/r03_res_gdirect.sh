#!/bin/bash
# resident kernel: gradient operands as direct element loads (DIMN_RES_GDIRECT=1) vs through wave-private LDS staging; one box, K = 5 share
cd "$(dirname "$0")/.." || exit 1
bash tools/ab_lib.sh 'for a in "" "--precision bf16"; do timeout 300 python bench.py --limit-subnets 5 --epochs 6 $a --no-cpu-baseline --steps 1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"lane_step_us %.2f  val %.6f\" % (1e3*d[\"config\"][\"lane_step_ms\"], d[\"config\"][\"final_val_loss\"]))"; done' "-DDIMN_RES_GDIRECT=0" "-DDIMN_RES_GDIRECT=1"
rm -f deepimpute_amd/csrc/libdimn_ab*.so
