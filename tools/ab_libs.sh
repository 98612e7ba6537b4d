#!/bin/bash
# same-box A/B of BUILDS of libdimn on one rank's share of the 8-GPU job (5 sub-nets, the register-resident epoch kernel):
#   tools/ab_libs.sh libdimn_x.so libdimn_y.so ...   (names under deepimpute_amd/csrc; env REPS, default 2; BENCH_ARGS)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ablibs; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-accuracy ${BENCH_ARGS:---limit-subnets 5}"
for rep in $(seq 1 ${REPS:-2}); do for v in "$@"; do
  DIMN_LIB_PATH=$PWD/deepimpute_amd/csrc/$v $B > $O/$v.$rep.json 2> $O/$v.$rep.err
  python -c "import json; d=json.load(open('$O/$v.$rep.json')); print('%-22s rep %d  %8.0f cells/s  %.3f us/step  val %.9f' % ('$v', $rep, d['value'], 1e3*d['config']['lane_step_ms'], d['config']['final_val_loss']))" || tail -5 $O/$v.$rep.err
done; done
