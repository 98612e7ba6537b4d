#!/bin/bash
# Round-6 evidence for the general path (one box): the shape families on the shipped build and with the batch-row GEMMs forced back onto the
# LDS-staged k_gen_gemm (DIMN_RES_TEST=gemm=0), per-kernel statistics of a general-path epoch, and the code-size probe of k_gen_rowgemm.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
bash tools/gen_ab.sh > $O/ab_stdout.txt 2>&1
cp gpurun_out/genab/ab.txt $O/general_ab.txt
cp gpurun_out/genab/kernel_stats_general.txt $O/kernel_stats_general.txt
cp gpurun_out/genab/kernel_stats_general_b128.txt $O/kernel_stats_general_b128.txt
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy"
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
run fam_h256 "--epochs 4"
run fam_h256_general "--general --epochs 4"
run fam_h512_general "--hidden 512 --epochs 4"
run fam_b128_general "--batch 128 --epochs 4"
run fam_h512_b128_general "--batch 128 --hidden 512 --epochs 4"
} | tee $O/families_general.txt
if [ -x tools/probe/gemm_probe ]; then
  { for a in "40 16 512 1 64" "40 256 512 1 64" "40 2400 256 1 64" "40 2400 256 3 64" "40 2400 256 4 64" "1 2400 64 1 64"; do tools/probe/gemm_probe $a; done; } > $O/gemm_probe.txt 2>&1
  cat $O/gemm_probe.txt
fi
