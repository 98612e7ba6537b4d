#!/bin/bash
# k_mid_pipe against k_mid_fused: the standalone probe (outputs compared, warm / cold timings, stamps), the parity tests that cover the
# fused second layer, and the step time of the cfg3 job with either kernel (same box, alternating).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/midp; mkdir -p $O
hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/k_probe_mid tools/k_probe_mid.hip 2> $O/probe_build.err && timeout 300 /tmp/k_probe_mid 40 > $O/probe.txt 2>&1
grep -i "pipe\|fused" $O/probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k second_layer > $O/tests.log 2>&1; tail -4 $O/tests.log
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy --epochs 4"
for i in 1 2; do
  for v in 0 1; do
    DIMN_MID_PIPE=$v timeout 600 $B > $O/bench_pipe${v}_$i.json 2> $O/bench_pipe${v}_$i.err
    python - <<PY
import json
d=json.loads(open("$O/bench_pipe${v}_$i.json").read().strip().splitlines()[-1]); c=d["config"]
print("DIMN_MID_PIPE=$v run $i: %.0f cells/s  lane_step %.4f ms  step wall %.4f ms  val %.5f" % (d["value"], c["lane_step_ms"], c["train_step_ms_wall"], c["final_val_loss"]))
PY
  done
done
