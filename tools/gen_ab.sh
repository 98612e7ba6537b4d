#!/bin/bash
# General-path A/B on one box: k_gen_rowgemm (default) against the LDS-staged k_gen_gemm (DIMN_RES_TEST=gemm=0) at the three shape families.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/genab; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy"
run() { # name env args
  env $2 timeout 600 $B $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$1.json")); c=d["config"]; print("%-34s %8.0f cells/s  step wall %.4f ms  lane_step %.4f ms  val %.4f" % ("$1", d["value"], c["train_step_ms_wall"], c["lane_step_ms"], c["final_val_loss"]))
except Exception as e: print("$1 failed", e, open("$O/$1.err").read()[-400:])
PY
}
{
for rep in 1 2; do
run fam_h256_general_new X=1 "--general --epochs 4"
run fam_h256_general_old DIMN_RES_TEST=gemm=0 "--general --epochs 4"
done
run fam_b128_general_new X=1 "--batch 128 --epochs 4"
run fam_b128_general_old DIMN_RES_TEST=gemm=0 "--batch 128 --epochs 4"
run fam_h512_b128_general_new X=1 "--batch 128 --hidden 512 --epochs 4"
run fam_h512_b128_general_old DIMN_RES_TEST=gemm=0 "--batch 128 --hidden 512 --epochs 4"
for s in ${GS:-}; do run fam_h256_general_gs$s DIMN_RES_TEST=gs=$s "--general --epochs 4"; done
} | tee $O/ab.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_gen -o run -- $B --general --epochs 1 --warmup 0 > $O/bench_general_under_rocprof.json 2>> $O/prof.err
python tools/kstats.py $O/prof_gen > $O/kernel_stats_general.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_gen128 -o run -- $B --batch 128 --epochs 1 --warmup 0 > $O/bench_general128_under_rocprof.json 2>> $O/prof.err
python tools/kstats.py $O/prof_gen128 > $O/kernel_stats_general_b128.txt 2>&1
rm -rf $O/prof_gen $O/prof_gen128
head -14 $O/kernel_stats_general.txt; head -14 $O/kernel_stats_general_b128.txt
