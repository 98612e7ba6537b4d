#!/bin/bash
# PMC passes of round 3 (separate passes, csv output): HBM traffic and MFMA utilisation per kernel at cfg3, and the memory-side traffic
# of the resident kernel at the 8-GPU share.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --epochs 1"
for c in FETCH_SIZE WRITE_SIZE; do timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- $B > /dev/null 2> $O/pmc_$c.err; done
python tools/pmc_traffic.py $O > $O/traffic.json 2> $O/pmc.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o pmc -- $B > /dev/null 2> $O/pmc_mfma.err
python tools/pmc_mfma.py $O/pmc_mfma > $O/mfma_util.json 2>> $O/pmc.err
for c in FETCH_SIZE WRITE_SIZE; do timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/k5/pmc_$c -o pmc -- $B --limit-subnets 5 --epochs 2 > /dev/null 2>> $O/pmc.err; done
python tools/pmc_traffic.py $O/k5 > $O/traffic_k5.json 2>> $O/pmc.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma $O/k5
python - <<PY
import json
for f in ("traffic","mfma_util","traffic_k5"):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, json.dumps(d.get("kernels", d))[:1500])
    except Exception as e: print(f, "unreadable", e)
PY
