#!/bin/bash
# End-of-round rocprofv3 --kernel-trace --stats summaries with the final code: the default bench command and one rank of the 8-GPU job.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03final; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- $B > $O/bench_under_rocprof.json 2> $O/prof.err
python tools/kstats.py $O/prof > $O/kernel_stats.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_k5 -o run -- $B --limit-subnets 5 --epochs 4 > $O/bench_k5_under_rocprof.json 2>> $O/prof.err
python tools/kstats.py $O/prof_k5 > $O/kernel_stats_k5.txt 2>&1
rm -rf $O/prof $O/prof_k5
head -8 $O/kernel_stats.txt; head -5 $O/kernel_stats_k5.txt
python - <<PY
import json
for f in ("bench_under_rocprof","bench_k5_under_rocprof"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["config"]["lane_step_ms"], d["roofline"]["avg_launch_ms"])
PY
