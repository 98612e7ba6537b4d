cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/c2
timeout 1500 python -m pytest tests/test_gpu_multinet.py -m gpu -q -x -s -k "cfg2_full_size" 2>&1 | tail -40
timeout 900 python bench.py --steps 1 --warmup 0 --no-dropin --cpu-budget 4 2> gpurun_out/c2/acc.err > gpurun_out/c2/acc.json; python -c "
import json
d=json.load(open('gpurun_out/c2/acc.json')); print(json.dumps(d['accuracy'], indent=1))"
