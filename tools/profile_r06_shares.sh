#!/bin/bash
# the per-rank shares of the N-GPU job again (after the last resident-kernel changes of round 6): K = 5 under rocprofv3, K = 5 / 10 / 20, configs[1], bf16
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06s; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_k5 -o run -- $B --limit-subnets 5 --epochs 4 > $O/bench_k5_under_rocprof.json 2>> $O/prof.err
python tools/kstats.py $O/prof_k5 > $O/kernel_stats_k5.txt 2>&1; rm -rf $O/prof_k5; head -5 $O/kernel_stats_k5.txt
run() { # name args
  timeout 900 $B $2 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$1.json")); c=d["config"]; print("%-26s %8.0f cells/s  step wall %.4f ms  lane_step %.4f ms  frac %.3f  val %.4f" % ("$1", d["value"], c["train_step_ms_wall"], c["lane_step_ms"], d["roofline"].get("frac") or 0, c["final_val_loss"]))
except Exception as e: print("$1 failed", e, open("$O/$1.err").read()[-400:])
PY
}
{
run bench_k5_resident "--limit-subnets 5"
DIMN_RESIDENT=0 run bench_k5_streaming "--limit-subnets 5"
run bench_k10 "--limit-subnets 10"
run bench_k20 "--limit-subnets 20"
run bench_k5_bf16_e6 "--limit-subnets 5 --epochs 6 --precision bf16"
run bench_cfg2 "--config cfg2"
} | tee $O/families_shares.txt
