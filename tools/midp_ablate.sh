#!/bin/bash
# k_mid_pipe: standalone probe with parts of the kernel removed (tools/k_probe_mid.hip, DIMN_MIDP_ABL bits) -- what a tile block waits for
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/midp; mkdir -p $O
for a in ${ABLS:-0 1 2 6 8 16 31}; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDIMN_MIDP_ABL=$a -o /tmp/k_probe_mid_$a tools/k_probe_mid.hip 2> $O/probe_build_$a.err && timeout 300 /tmp/k_probe_mid_$a 40 > $O/probe_abl$a.txt 2>&1
  echo "== ablation $a"; grep "k_mid_pipe (warm)\|cold (after 1 GB of other traffic): k_mid_pipe\|pipe blocks\|pipe timeline" $O/probe_abl$a.txt
done
