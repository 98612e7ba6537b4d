#!/bin/bash
# PMC passes of round 2 (separate passes, csv output): HBM traffic and MFMA utilisation per kernel at cfg3.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --epochs 1"
for c in FETCH_SIZE WRITE_SIZE; do timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- $B > /dev/null 2> $O/pmc_$c.err; done
python tools/pmc_traffic.py $O > $O/traffic.json 2> $O/pmc.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o pmc -- $B > /dev/null 2> $O/pmc_mfma.err
python tools/pmc_mfma.py $O/pmc_mfma > $O/mfma_util.json 2>> $O/pmc.err
# the 8-GPU share (register-resident kernel): its HBM traffic per epoch launch
for c in FETCH_SIZE WRITE_SIZE; do timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/k5/pmc_$c -o pmc -- $B --limit-subnets 5 --epochs 2 > /dev/null 2>> $O/pmc.err; done
python tools/pmc_traffic.py $O/k5 > $O/traffic_k5.json 2>> $O/pmc.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin > /dev/null 2>> $O/pmc.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv
rm -rf $O/prof $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma $O/k5
cat $O/traffic.json | head -40; cat $O/mfma_util.json | head -30; cat $O/traffic_k5.json | head -20
