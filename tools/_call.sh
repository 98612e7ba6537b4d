cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/c6
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/c6/tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c6/tests.log | tail -5; grep -B30 "short test summary" gpurun_out/c6/tests.log | head -60
for rep in 1 2; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-accuracy > gpurun_out/c6/bench_dropin_$rep.json 2> gpurun_out/c6/bench_dropin_$rep.err
python - <<PY
import json
d=json.load(open("gpurun_out/c6/bench_dropin_$rep.json")); print(d["value"], json.dumps(d["config"]["dropin"]))
PY
done
