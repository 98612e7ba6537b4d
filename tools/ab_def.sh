#!/bin/bash
# A/B of compile-time switches on one box: tools/ab_def.sh "-DA=1" "-DA=0" ...  (each argument = one libdimn build)
cd deepimpute_amd/csrc
cp libdimn.so libdimn_product.so
i=0; for f in "$@"; do hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 $f -o libdimn_v$i.so dimn.hip -ldl 2>/dev/null; i=$((i+1)); done
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-18s cells/s %.0f  step_ms %.4f  w1_launch_ms %.4f  frac %.3f  val %.6f' % ('$1', d['value'], d['config']['lane_step_ms'], r['avg_launch_ms'], r['frac'], d['config']['final_val_loss']))"; }
for rep in 1 2; do i=0; for f in "$@"; do cp libdimn_v$i.so libdimn.so; (cd ../.. && python bench.py --no-cpu-baseline 2>/dev/null | show "$f"); i=$((i+1)); done; done
cp libdimn_product.so libdimn.so
