#!/bin/bash
# in-step duration of the second-layer kernel on a handle of precision bf16: k_mid_pipe<BF> against k_mid_fused<KEEP, BF> (rocprofv3 kernel trace, 2 epochs of cfg3)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/midp; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-accuracy --epochs 2 --precision bf16"
for v in 1 0; do
  rm -rf $O/profb_$v
  DIMN_MID_PIPE=$v timeout 600 rocprofv3 --kernel-trace --stats -d $O/profb_$v -o run -- $B > $O/instep_bf16_$v.json 2> $O/instep_bf16_$v.err
  python tools/kstats.py $O/profb_$v > $O/instep_bf16_kstats_$v.txt 2>&1
  echo "== DIMN_MID_PIPE=$v"; head -6 $O/instep_bf16_kstats_$v.txt
  rm -rf $O/profb_$v
done
