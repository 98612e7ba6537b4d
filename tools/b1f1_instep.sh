#!/bin/bash
# in-step duration of every kernel of the cfg3 step (rocprofv3 kernel trace, 2 epochs), library under test vs DIMN_LIB_PATH_B (if set)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/midp; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-accuracy --epochs 2"
for rep in 1 2; do for v in A B; do
  if [ $v = B ]; then [ -z "$DIMN_LIB_PATH_B" ] && continue; E="DIMN_LIB_PATH=$DIMN_LIB_PATH_B"; else E="X=1"; fi
  rm -rf $O/profx
  env $E timeout 600 rocprofv3 --kernel-trace --stats -d $O/profx -o run -- $B > $O/instepx.json 2> $O/instepx.err
  python tools/kstats.py $O/profx > $O/instepx_kstats.txt 2>&1
  echo "== $v rep $rep"; sed -n 2,3p $O/instepx_kstats.txt; sed -n 5,6p $O/instepx_kstats.txt | grep reduce
  rm -rf $O/profx
done; done
