#!/bin/bash
# A/B on one box of the DIMN_NT cache-policy switch (default 18: non-temporal state stores + m,v loads; 2: stores only; 0: plain); NTS="18 2 0"
cd deepimpute_amd/csrc
cp libdimn.so libdimn_product.so
for v in ${NTS:-18 2 0}; do hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -DDIMN_NT=$v -o libdimn_nt$v.so dimn.hip -ldl 2>/dev/null; done
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1  cells/s %.0f  step_ms %.4f  w1_launch_ms %.4f  frac %.3f  val %.6f' % (d['value'], d['config']['lane_step_ms'], r['avg_launch_ms'], r['frac'], d['config']['final_val_loss']))"; }
for rep in 1 2; do for v in ${NTS:-18 2 0}; do cp libdimn_nt$v.so libdimn.so; (cd ../.. && python bench.py --no-cpu-baseline 2>/dev/null | show "NT=$v"); done; done
cp libdimn_product.so libdimn.so
