cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/c7
( timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/c7/tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c7/tests.log | tail -5; grep -B30 "short test summary" gpurun_out/c7/tests.log | head -50
bash tools/cli_cold.sh both > gpurun_out/c7/cli_default.txt 2>&1; grep -v "^    " gpurun_out/c7/cli_default.txt | tail -32
bash tools/cli_cold.sh big --hidden-neurons 256 --max-epochs 18 > gpurun_out/c7/cli_h256_e18.txt 2>&1; tail -44 gpurun_out/c7/cli_h256_e18.txt
