#!/bin/bash
# round 4, GPU call 3: per-kernel times of the general path (rocprofv3 kernel trace), the drop-in stages after the round's host changes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_general -- python bench.py --general --epochs 1 --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-accuracy > $O/general.json 2> $O/general.err
python tools/kstats.py $O/prof_general > $O/kstats_general.txt 2>&1; head -20 $O/kstats_general.txt
rocprofv3 --kernel-trace --stats -d $O/prof_h300 -- python bench.py --hidden 300 --epochs 1 --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-accuracy > $O/h300.json 2> $O/h300.err
python tools/kstats.py $O/prof_h300 > $O/kstats_h300.txt 2>&1; head -14 $O/kstats_h300.txt
timeout 600 python tools/dropin_stages.py 50000 20000 18 > $O/dropin_stages.txt 2>&1; grep -v "^Net \|genes selected\|^\[" $O/dropin_stages.txt | tail -60
rm -rf $O/prof_general/*/*.db $O/prof_h300/*/*.db 2>/dev/null
du -sh $O
