#!/bin/bash
# BASELINE configs[4], one rank's true share: 8 of the 59 sub-nets at the FULL 1 000 000 cells x 30 000 genes, bf16, matrix streamed from host
# memory (120 GB) every impute; under rocprofv3 --kernel-trace --stats (a handful of launches per epoch on the resident kernel)
cd "$(dirname "$0")/.." || exit 1
O=$PWD/gpurun_out/r06cfg5; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 2400 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --limit-subnets 8 --precision bf16 --stream --steps 1 --warmup 0 --no-cpu-baseline --no-dropin > $O/bench_cfg5_1M.json 2> $O/bench.err
tail -3 $O/bench.err
python $GRAFT_REPO_ROOT/tools/kstats.py $O/prof > $O/kernel_stats_cfg5_1M.txt 2>&1
head -12 $O/kernel_stats_cfg5_1M.txt
python - <<PY
import json
d=json.load(open("$O/bench_cfg5_1M.json")); c=d["config"]
print("value", d["value"], "ms/impute", d["ms_per_step"], "lane_step_ms", c["lane_step_ms"], "handover", c.get("streamed_handover"), "predict", d["roofline"].get("predict"), "synth s", c.get("synth_seconds"))
PY
rm -rf $O/prof
