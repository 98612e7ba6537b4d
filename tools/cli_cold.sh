#!/bin/bash
# the deepImpute CLI as ONE COLD PROCESS (reference deepimpute/deepImpute.py:6-40: read CSV -> fit -> predict -> write CSV)
# at configs[1] (5k x 5k) and configs[2] (50k x 20k) size, wall time by stage.  tools/cli_cold.sh [small|big|both] [extra CLI flags]
# The synthetic matrix is the bench's (BASELINE.md generator), written as the integer count CSV the tool is specified for.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/cli; mkdir -p $O
which=${1:-both}; shift
g++ -O2 -o /tmp/counts_to_csv tools/counts_to_csv.cpp || exit 1
run() { # name cells genes [flags]
  name=$1; n=$2; g=$3; shift 3
  csv=/tmp/$name.csv
  if [ ! -f $csv ]; then
    python - <<PY
import numpy as np, sys
sys.path.insert(0, ".")
import bench
norm = bench.synth_counts($n, $g, seed=0)
np.rint(np.expm1(norm.astype(np.float64))).astype(np.int32).tofile("/tmp/$name.bin")
PY
    /tmp/counts_to_csv /tmp/$name.bin $n $g $csv && rm -f /tmp/$name.bin
  fi
  ls -la $csv | awk '{print "input:", $5/1e6, "MB"}'
  sync; echo 3 > /proc/sys/vm/drop_caches 2>/dev/null      # a cold file system where the box lets us
  t0=$(date +%s.%N)
  DIMN_TRACE=1 python -m deepimpute_amd.deepImpute $csv -o /tmp/$name.out.csv --limit $g "$@" > $O/$name.out 2> $O/$name.err
  rc=$?
  t1=$(date +%s.%N)
  python - <<PY
import re
t0, t1 = $t0, $t1
err = open("$O/$name.err").read()
m = re.search(r"\[deepImpute\] t0 ([0-9.]+)", err)
print("== $name ($n x $g) $*: rc $rc, process wall %.2f s" % (t1 - t0))
if m:
    print("  interpreter start -> package import  %8.3f s" % (float(m.group(1)) - t0))
for line in err.splitlines():
    if line.startswith("[deepImpute] ") and " t0 " not in line:
        print(" ", line[len("[deepImpute] "):])
for line in open("$O/$name.out").read().splitlines():
    if line.startswith("Stopped fitting") or line.startswith("[{'type'"):
        print(" ", line)
PY
  ls -la /tmp/$name.out.csv | awk '{print "  output:", $5/1e6, "MB"}'
  rm -f /tmp/$name.out.csv
}
if [ $which = small ] || [ $which = both ]; then run cfg2 5000 5000 "$@"; run cfg2 5000 5000 "$@"; fi
if [ $which = big ] || [ $which = both ]; then run cfg3 50000 20000 "$@"; fi
