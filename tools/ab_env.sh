#!/bin/bash
# same-box A/B of library switches inside the cfg3 step: tools/ab_env.sh "DIMN_X=0" "DIMN_X=1" ["..."]; env BENCH_ARGS (default: 4 epochs), REPS (default 2)
# prints per run: cells/s, lane_step_ms (HIP events around the step), the dominant kernel's stamped launch time, final validation loss
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ab; mkdir -p $O
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy ${BENCH_ARGS:---epochs 4}"
for rep in $(seq 1 ${REPS:-2}); do for v in "$@"; do
  tag=$(echo "$v" | tr ' =/' '___')
  env $v timeout 900 $B > $O/$tag.$rep.json 2> $O/$tag.$rep.err
  python - "$O/$tag.$rep.json" "$v" $rep <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]; r = d["roofline"]
    print("%-34s rep %s  %8.0f cells/s  lane_step %.4f ms  kernel %.2f us  frac %.3f  val %.5f" % (sys.argv[2], sys.argv[3], d["value"], c["lane_step_ms"], 1e3 * r.get("avg_launch_ms", 0), r.get("frac") or 0, c["final_val_loss"]))
except Exception as e:
    print(sys.argv[2], "failed:", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
done; done
