#!/bin/bash
# in-step duration of the second-layer kernel (rocprofv3 kernel trace of a 2-epoch cfg3 run), k_mid_pipe and k_mid_fused
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/midp; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-accuracy --epochs 2"
for v in ${VARIANTS:-1 0}; do
  rm -rf $O/prof_$v
  DIMN_MID_PIPE=$v timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o run -- $B > $O/instep_$v.json 2> $O/instep_$v.err
  python tools/kstats.py $O/prof_$v > $O/instep_kstats_$v.txt 2>&1
  echo "== DIMN_MID_PIPE=$v"; head -8 $O/instep_kstats_$v.txt
  rm -rf $O/prof_$v
done
