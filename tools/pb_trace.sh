#!/bin/bash
# Per-workgroup phase stamps of k_predict_bf16 (diagnostic build of the library; run on the GPU box from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pb_trace.sh > gpurun_out/pb_trace.txt 2>&1'
set -e
cd "$(dirname "$0")/.."
( cd deepimpute_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -march=x86-64-v3 -Wno-unused-result -DDIMN_PB_TRACE=1 $PB_TRACE_DEFS -o /tmp/libdimn_trace.so dimn.hip -ldl -lpthread )
DIMN_LIB_PATH=/tmp/libdimn_trace.so DIMN_PREDICT_TRACE=/tmp/pb_trace.bin python bench.py --precision bf16 --epochs 1 --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-accuracy > /tmp/pb_bench.json
python tools/pb_trace.py /tmp/pb_trace.bin
