#!/bin/bash
# Where the tile loop of the resident kernel spends its time: builds with parts of it removed (WRONG results, timing only), K = 5 share
cd "$(dirname "$0")/.." || exit 1
bash tools/ab_lib.sh 'timeout 300 python bench.py --limit-subnets 5 --no-cpu-baseline --epochs 4 --steps 1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"lane_step_us %.2f  kernel %s\" % (1e3*d[\"config\"][\"lane_step_ms\"], d[\"roofline\"][\"kernel\"][:16]))"' "-DDIMN_RES_ABL=0" "-DDIMN_RES_ABL=2" "-DDIMN_RES_ABL=4" "-DDIMN_RES_ABL=6"
rm -f deepimpute_amd/csrc/libdimn_ab*.so
