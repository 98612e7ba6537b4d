#!/bin/bash
# Round-3 evidence on one box: the default bench line, the same command under rocprofv3 (kernel stats), the 8- / 4- / 2-GPU shares
# (K = 5 / 10 / 20: register-resident epoch kernel, manager protocol) with the round-2 library (tools/libdimn_old.so, if present)
# beside them, cfg2, cfg3 at --precision bf16, and the resident kernel's phase timeline.  Output under gpurun_out/r03/.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- $B > $O/bench_under_rocprof.json 2> $O/prof.err
python tools/kstats.py $O/prof > $O/kernel_stats.txt 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
# one rank of the 8-GPU job: 5 sub-nets, resident kernel; the streaming kernels and the round-2 kernel on the same box
timeout 400 python bench.py --limit-subnets 5 --no-cpu-baseline > $O/bench_k5_resident.json 2>> $O/bench.err
DIMN_RESIDENT=0 timeout 400 python bench.py --limit-subnets 5 --no-cpu-baseline > $O/bench_k5_streaming.json 2>> $O/bench.err
if [ -f tools/libdimn_old.so ]; then DIMN_LIB_PATH=tools/libdimn_old.so timeout 400 python bench.py --limit-subnets 5 --no-cpu-baseline > $O/bench_k5_resident_r02kernel.json 2>> $O/bench.err; fi
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_k5 -o run -- $B --limit-subnets 5 --epochs 4 > /dev/null 2>> $O/prof.err
python tools/kstats.py $O/prof_k5 > $O/kernel_stats_k5.txt 2>&1
for k in 10 20; do timeout 400 python bench.py --limit-subnets $k --no-cpu-baseline --epochs 6 --steps 1 > $O/bench_k$k.json 2>> $O/bench.err; done
DIMN_RESIDENT=0 timeout 400 python bench.py --limit-subnets 10 --no-cpu-baseline --epochs 6 --steps 1 > $O/bench_k10_streaming.json 2>> $O/bench.err
DIMN_RESIDENT=0 timeout 400 python bench.py --limit-subnets 20 --no-cpu-baseline --epochs 6 --steps 1 > $O/bench_k20_streaming.json 2>> $O/bench.err
timeout 300 python bench.py --config cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2>> $O/bench.err
timeout 300 $B --epochs 6 > $O/bench_cfg3_f32_e6.json 2>> $O/bench.err
timeout 300 $B --epochs 6 --precision bf16 > $O/bench_cfg3_bf16_e6.json 2>> $O/bench.err
timeout 300 $B --epochs 6 --precision bf16 --limit-subnets 5 > $O/bench_k5_bf16_e6.json 2>> $O/bench.err
timeout 600 python tools/res_timeline.py 5 > $O/resident_timeline.txt 2>> $O/bench.err
rm -rf $O/prof $O/prof_k5
for f in $O/bench*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); c=d["config"]; print("%-46s %9.0f cells/s  lane_step %.4f ms  frac %.3f  dropin %s" % ("$f".split("/")[-1], d["value"], c["lane_step_ms"], d["roofline"]["frac"], (c.get("dropin") or {}).get("cells_per_s")))
except Exception as e: print("$f", "unreadable", e)
PY
done
