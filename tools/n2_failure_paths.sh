#!/bin/bash
# N = 2 on a ONE-GPU box: both failure classes must end in the contract's line with value null on rank 0 and a non-zero exit
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04n2; mkdir -p $O
echo "== rank 1 has no device (dimn_create fails before any collective)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 0 --epochs 1 --config cfg2 > $O/nodevice.out 2> $O/nodevice.err; echo "rc=$?"
grep -o '"value": [a-z0-9.]*\|"rccl_error": "[^"]*"' $O/nodevice.out | head -3; grep -c Traceback $O/nodevice.err
echo "== both ranks on device 0 (RCCL refuses or hangs: bounded by the timeout)"
DIMN_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 0 --epochs 1 --config cfg2 > $O/samedevice.out 2> $O/samedevice.err; echo "rc=$?"
grep -o '"value": [a-z0-9.]*\|"rccl_error": "[^"]*"' $O/samedevice.out | head -3; tail -3 $O/samedevice.err | cut -c1-300
