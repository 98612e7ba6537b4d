#!/bin/bash
# resident kernel, M2: 8 vs 16 dD requests in flight per thread (one box, K = 5 share, 6 epochs)
cd "$(dirname "$0")/.." || exit 1
bash tools/ab_lib.sh 'python bench.py --limit-subnets 5 --no-cpu-baseline --epochs 6 --steps 2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"lane_step_us %.2f  val %.6f\" % (1e3*d[\"config\"][\"lane_step_ms\"], d[\"config\"][\"final_val_loss\"]))"' "-DDIMN_RES_M2WIN=8" "-DDIMN_RES_M2WIN=16"
rm -f deepimpute_amd/csrc/libdimn_ab*.so
