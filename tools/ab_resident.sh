#!/bin/bash
# A/B of the resident epoch kernel's compile-time variants on one box (phase timeline per variant).
#   tools/ab_resident.sh "<flags of variant 1>" "<flags of variant 2>" ...
mkdir -p gpurun_out
i=0
for flags in "$@"; do
  i=$((i+1))
  echo "=== variant $i: $flags" | tee -a gpurun_out/ab_resident.log
  timeout 600 python tools/res_timeline.py 5 $flags 2>&1 | tail -34 | tee -a gpurun_out/ab_resident.log
done
