#!/bin/bash
# in-step duration of the second-layer kernel (rocprofv3 kernel trace of a 2-epoch cfg3 run), k_mid_pipe and k_mid_fused
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/midp; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-accuracy --epochs 2"
for v in ${VARIANTS:-1 0}; do   # 1: k_mid_pipe (default) / 0: k_mid_fused
  case $v in 1) E="DIMN_MID_PIPE=1";; 0) E="DIMN_MID_PIPE=0";; esac
  rm -rf $O/prof_$v
  env $E timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o run -- $B > $O/instep_$v.json 2> $O/instep_$v.err
  python tools/kstats.py $O/prof_$v > $O/instep_kstats_$v.txt 2>&1
  echo "== variant $v ($E)"; head -8 $O/instep_kstats_$v.txt; python -c "
import json; d=json.loads(open('$O/instep_$v.json').read().strip().splitlines()[-1]); print('lane_step_ms', d['config']['lane_step_ms'], 'val', d['config']['final_val_loss'])"
  rm -rf $O/prof_$v
done
