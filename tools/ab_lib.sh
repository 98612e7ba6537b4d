#!/bin/bash
# A/B of compile-time variants of libdimn on one box: builds one library per flag set and runs a command with each.
#   tools/ab_lib.sh "<command>" "<flags of variant 1>" "<flags of variant 2>" ...
# The command sees DIMN_LIB_PATH; its stdout lines are tagged with the variant in gpurun_out/ab_lib.log.
cmd="$1"; shift
mkdir -p gpurun_out
i=0
for flags in "$@"; do
  i=$((i+1))
  lib=deepimpute_amd/csrc/libdimn_ab$i.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 $flags -o $lib deepimpute_amd/csrc/dimn.hip -ldl -lpthread || exit 1
done
for rep in 1 2; do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    echo "=== variant $i rep $rep: $flags" | tee -a gpurun_out/ab_lib.log
    DIMN_LIB_PATH=$PWD/deepimpute_amd/csrc/libdimn_ab$i.so bash -c "$cmd" 2>/dev/null | tee -a gpurun_out/ab_lib.log
  done
done
