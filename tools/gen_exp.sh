#!/bin/bash
# A general-path epoch under rocprofv3 --kernel-trace --stats once per DIMN_RES_TEST value given (e.g. `tools/gen_exp.sh x=1 gemm=0 gs=2 gfuse=0`):
# the step time and the per-kernel table of each -- how the round-6 variants of the batch-row GEMMs were compared kernel by kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/genexp; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy"
for v in "$@"; do
  tag=$(echo $v | tr '=,' '__')
  DIMN_RES_TEST=$v timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o run -- $B --general --epochs 1 --warmup 0 > $O/b_$tag.json 2>> $O/prof.err
  python tools/kstats.py $O/prof_$tag > $O/ks_$tag.txt 2>&1
  rm -rf $O/prof_$tag
  echo "== $v: $(python -c "import json;d=json.load(open('$O/b_$tag.json'));print(d['config']['train_step_ms_wall'])")"; head -9 $O/ks_$tag.txt
done
