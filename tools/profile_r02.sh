#!/bin/bash
# Round-2 evidence on one box: the default bench line, the same command under rocprofv3 (kernel stats), PMC traffic / MFMA
# passes, and the 8-GPU share (K = 5, register-resident epoch kernel) with its kernel stats.  Output under gpurun_out/r02/.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- $B > $O/bench_under_rocprof.json 2> $O/prof.err
python tools/kstats.py $O/prof > $O/kernel_stats.txt 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do timeout 900 rocprofv3 --pmc $c -d $O/pmc_$c -o run -- $B --epochs 2 > /dev/null 2> $O/pmc_$c.err; done
python tools/pmc_traffic.py $O > $O/traffic.json 2>> $O/prof.err
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o run -- $B --epochs 2 > /dev/null 2> $O/pmc_mfma.err
python tools/pmc_mfma.py $O/pmc_mfma > $O/mfma_util.json 2>> $O/prof.err
# one rank of the 8-GPU job: 5 sub-nets, resident kernel; and the streaming kernels on the same box
timeout 400 python bench.py --limit-subnets 5 --no-cpu-baseline > $O/bench_k5_resident.json 2>> $O/bench.err
DIMN_RESIDENT=0 timeout 400 python bench.py --limit-subnets 5 --no-cpu-baseline > $O/bench_k5_streaming.json 2>> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_k5 -o run -- $B --limit-subnets 5 --epochs 4 > /dev/null 2>> $O/prof.err
python tools/kstats.py $O/prof_k5 > $O/kernel_stats_k5.txt 2>&1
for k in 10 20; do timeout 400 python bench.py --limit-subnets $k --no-cpu-baseline --epochs 6 --steps 1 > $O/bench_k$k.json 2>> $O/bench.err; done
DIMN_RESIDENT=0 timeout 400 python bench.py --limit-subnets 10 --no-cpu-baseline --epochs 6 --steps 1 > $O/bench_k10_streaming.json 2>> $O/bench.err
timeout 300 python bench.py --config cfg2 > $O/bench_cfg2.json 2>> $O/bench.err
rm -rf $O/prof $O/prof_k5 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
ls -la $O
