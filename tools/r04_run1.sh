#!/bin/bash
# round 4, GPU call 1: the -m gpu suite, then the driver's bench command with and without the arena cache (hand-over traces on stderr)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r04_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04_tests.log
DIMN_TRACE=1 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driverlike.json 2> gpurun_out/r04_bench_driverlike.err
echo "bench rc=$?"
DIMN_ARENA_CACHE_GB=0 DIMN_TRACE=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy > gpurun_out/r04_bench_nocache.json 2> gpurun_out/r04_bench_nocache.err
echo "bench(nocache) rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r04_bench_driverlike.json", "gpurun_out/r04_bench_nocache.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["config"]["lane_step_ms"], json.dumps(d["config"].get("dropin"))[:1500])
    except Exception as e:
        print(f, "unreadable", e)
PY
grep -h "gather:\|set_matrix_counts" gpurun_out/r04_bench_driverlike.err | tail -8
grep -h "gather:\|set_matrix_counts" gpurun_out/r04_bench_nocache.err | tail -8
