#!/bin/bash
# k_predict_bf16 ablations (what could the output stores / the X stream give back at most?).  Diagnostic builds of the library
# (wrong results by construction), made on the build host BEFORE the call:  tools/pb_ablate.sh build ; then on the GPU box:  tools/pb_ablate.sh run
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build_abl
  for a in 1 2 3; do
    ( cd deepimpute_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -march=x86-64-v3 -Wno-unused-result -DDIMN_PB_ABL=$a -o ../../build_abl/libdimn_pb$a.so dimn.hip -ldl -lpthread ) &
  done
  wait; ls -la build_abl
  exit 0
fi
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
