#!/bin/bash
# End-of-round evidence of round 6 (one box): the driver's bench command, the same workload under rocprofv3 --kernel-trace --stats, one rank of
# the 8-GPU job (resident kernel), PMC passes (separate: traffic, MFMA), the shape families, configs[1], precision bf16.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06p; mkdir -p $O
export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- $B > $O/bench_under_rocprof.json 2> $O/prof.err
python tools/kstats.py $O/prof > $O/kernel_stats.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_k5 -o run -- $B --limit-subnets 5 --epochs 4 > $O/bench_k5_under_rocprof.json 2>> $O/prof.err
python tools/kstats.py $O/prof_k5 > $O/kernel_stats_k5.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_gen -o run -- $B --general --epochs 1 --warmup 0 > $O/bench_general_under_rocprof.json 2>> $O/prof.err
python tools/kstats.py $O/prof_gen > $O/kernel_stats_general.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_bf -o run -- $B --precision bf16 --epochs 2 > $O/bench_bf16_under_rocprof.json 2>> $O/prof.err
python tools/kstats.py $O/prof_bf > $O/kernel_stats_bf16.txt 2>&1
rm -rf $O/prof $O/prof_k5 $O/prof_gen $O/prof_bf
head -8 $O/kernel_stats.txt; head -5 $O/kernel_stats_k5.txt; head -10 $O/kernel_stats_general.txt
P="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-accuracy --epochs 1"
for c in FETCH_SIZE WRITE_SIZE; do timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- $P > /dev/null 2> $O/pmc_$c.err; done
python tools/pmc_traffic.py $O > $O/traffic.json 2> $O/pmc.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o pmc -- $P > /dev/null 2> $O/pmc_mfma.err
python tools/pmc_mfma.py $O/pmc_mfma > $O/mfma_util.json 2>> $O/pmc.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
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
run bench_cfg3_bf16_e6 "--epochs 6 --precision bf16"
run bench_cfg3_f32_e6 "--epochs 6"
run fam_h256 "--epochs 4"
run fam_h300 "--hidden 300 --epochs 4"
run fam_h256_general "--general --epochs 4"
run fam_b128_general "--batch 128 --epochs 4"
run fam_h512_b128_general "--batch 128 --hidden 512 --epochs 4"
} | tee $O/families.txt
bash tools/cli_cold.sh both > $O/cli_default.txt 2>&1
bash tools/cli_cold.sh big --hidden-neurons 256 --max-epochs 18 > $O/cli_h256_e18.txt 2>&1; grep -v "^    " $O/cli_h256_e18.txt | tail -12
python tools/finish_ab.py 2>&1 | grep -v "^\[dimn\]" | tail -8 | tee $O/finish_ab.txt
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print("BENCH", d["value"], d["ms_per_step"], d["config"]["lane_step_ms"], d["roofline"]["frac"], json.dumps(d["config"]["dropin"])[:900]); print(json.dumps(d["cpu_baseline"])[:300])
for f in ("traffic","mfma_util"):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, json.dumps(d.get("kernels", d))[:1200])
    except Exception as e: print(f, "unreadable", e)
PY
