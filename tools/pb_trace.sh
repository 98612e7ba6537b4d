#!/bin/bash
# k_predict_bf16 diagnostics (diagnostic builds of the library; from the repo root):
#   phase stamps per workgroup, on the GPU box:   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pb_trace.sh > gpurun_out/pb_trace.txt 2>&1'
#   ablations (what could the output stores / the X stream give back at most?; wrong results by construction):
#       tools/pb_trace.sh ablate-build      on the build host BEFORE the call (three libraries under build_abl/, -DDIMN_PB_ABL=1|2|3)
#       tools/pb_trace.sh ablate-run        on the GPU box  -> profiles/r05_predict_bf16_ablation.txt
set -e
cd "$(dirname "$0")/.."
HIPCC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -march=x86-64-v3 -Wno-unused-result"
if [ "$1" = ablate-build ]; then
  mkdir -p build_abl
  for a in 1 2 3; do ( cd deepimpute_amd/csrc && $HIPCC -DDIMN_PB_ABL=$a -o ../../build_abl/libdimn_pb$a.so dimn.hip -ldl -lpthread ) & done
  wait; ls -la build_abl
  exit 0
fi
if [ "$1" = ablate-run ]; then
  set +e
  O=gpurun_out/pb_abl; mkdir -p $O
  B="python bench.py --precision bf16 --epochs 1 --steps 2 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy"
  for rep in 1 2; do for v in base 1 2 3; do
    lib=deepimpute_amd/csrc/libdimn.so; [ $v != base ] && lib=build_abl/libdimn_pb$v.so
    DIMN_LIB_PATH=$PWD/$lib timeout 600 $B > $O/$v.$rep.json 2> $O/$v.$rep.err
    python - $O/$v.$rep.json $v $rep <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); p = d["roofline"]["predict"]
    print("abl %-5s rep %s  k_predict_bf16 %.3f ms  (%s)" % (sys.argv[2], sys.argv[3], p["ms"], {"base": "shipped kernel", "1": "no output stores", "2": "X rows from L2", "3": "neither"}[sys.argv[2]]))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
  done; done
  exit 0
fi
( cd deepimpute_amd/csrc && $HIPCC -DDIMN_PB_TRACE=1 $PB_TRACE_DEFS -o /tmp/libdimn_trace.so dimn.hip -ldl -lpthread )
DIMN_LIB_PATH=/tmp/libdimn_trace.so DIMN_PREDICT_TRACE=/tmp/pb_trace.bin python bench.py --precision bf16 --epochs 1 --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-accuracy > /tmp/pb_bench.json
python tools/pb_trace.py /tmp/pb_trace.bin
