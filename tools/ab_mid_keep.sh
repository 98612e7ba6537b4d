#!/bin/bash
# k_mid_fused with / without the W2 column blocks kept in LDS: kernel time (rocprofv3) and HBM traffic (PMC) per launch
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/keep; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --epochs 2"
for f in 1 0; do
  DIMN_MID_KEEP=$f timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof$f -o run -- $B > /dev/null 2> $O/prof$f.err
  echo "KEEP=$f"; python tools/kstats.py $O/prof$f | grep "k_mid_fused\|k_w1_update"
  mkdir -p $O/t$f
  for c in FETCH_SIZE WRITE_SIZE; do DIMN_MID_KEEP=$f timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/t$f/pmc_$c -o pmc -- $B --epochs 1 > /dev/null 2>> $O/pmc.err; done
  python tools/pmc_traffic.py $O/t$f | grep -A4 "k_mid_fused"
done
rm -rf $O/prof1 $O/prof0 $O/t1 $O/t0
