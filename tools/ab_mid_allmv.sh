#!/bin/bash
# k_mid_fused<KEEP> A/B on one box: kernel time under rocprofv3 and the bench step time, KEEP on / off
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/allmv; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --epochs 2"
for f in 1 0 1 0; do
  DIMN_MID_KEEP=$f timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- $B > /dev/null 2> $O/prof.err
  echo "KEEP=$f $(python tools/kstats.py $O/prof | grep "k_mid_fused" | awk '{print $1,$2,$3,"avg_us",$6}')"; rm -rf $O/prof
done
for f in 1 0 1 0; do echo "KEEP=$f $(DIMN_MID_KEEP=$f python bench.py --steps 2 --warmup 1 --no-dropin --epochs 2 2>/dev/null | python tools/predict_line.py)"; done
