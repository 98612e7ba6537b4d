#!/bin/bash
# round 4, GPU call 8: general path with the row gather fused into the first layer's GEMMs; the bench's drop-in leg with the finish pipeline's blocks made at warm-up; full -m gpu suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2700 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/tests.log | tail -2
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy"
run() { # name args
  timeout 900 $B $2 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$1.json")); c=d["config"]; print("%-22s %8.0f cells/s  step wall %.4f ms  lane_step %.4f ms  frac %.3f  val %.4f" % ("$1", d["value"], c["train_step_ms_wall"], c["lane_step_ms"], d["roofline"].get("frac") or 0, c["final_val_loss"]))
except Exception as e: print("$1 failed", e, open("$O/$1.err").read()[-400:])
PY
}
run h256 "--epochs 4"
run h256_general "--general --epochs 4"
run b128_general "--batch 128 --epochs 4"
run h512_b128_general "--batch 128 --hidden 512 --epochs 4"
run h300 "--hidden 300 --epochs 4"
rocprofv3 --kernel-trace --stats -d $O/prof_general -- python bench.py --general --epochs 1 --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-accuracy > $O/general.json 2> $O/general.err
python tools/kstats.py $O/prof_general > $O/kstats_general.txt 2>&1; head -12 $O/kstats_general.txt
for rep in 1 2; do
DIMN_TRACE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-accuracy > $O/bench_dropin_$rep.json 2> $O/bench_dropin_$rep.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04h/bench_dropin_$rep.json")); print(d["value"], json.dumps(d["config"]["dropin"]))
PY
grep "X / Y arenas\|finish:" $O/bench_dropin_$rep.err | tail -4
done
rm -rf $O/prof_general/*/*.db 2>/dev/null
