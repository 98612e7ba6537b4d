#!/bin/bash
# Kernel table of the drop-in MultiNet.fit + predict at 50k x 20k (2 epochs: the planning kernels are what this is for) under
# rocprofv3 --kernel-trace --stats, and the default bench line once more.  Output under gpurun_out/r03d/.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python tools/dropin_stages.py 50000 20000 2 > $O/dropin_stages_under_rocprof.txt 2> $O/prof.err
python tools/kstats.py $O/prof > $O/kernel_stats_dropin.txt 2>&1
rm -rf $O/prof
timeout 900 python tools/dropin_stages.py 50000 20000 18 > $O/dropin_stages.txt 2>> $O/prof.err
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
grep -v "^ *$" $O/kernel_stats_dropin.txt | head -40; tail -30 $O/dropin_stages.txt
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); c=d["config"]
print(d["value"], c["lane_step_ms"], d["roofline"]["frac"], c["dropin"]["fit_s"], c["dropin"]["predict_s"], c["dropin"]["cells_per_s"], c["dropin"]["stages_s"])
PY
