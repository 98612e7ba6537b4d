import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from deepimpute_amd._counts import DeviceCounts
from deepimpute_amd import _hostpar
n, g = 50000, 20000
counts = np.rint(np.expm1(bench.synth_counts(n, g, seed=0).astype(np.float64)))
for rep in range(3):
    t = time.perf_counter(); c = DeviceCounts.try_create(counts, 0); t1 = time.perf_counter() - t
    t = time.perf_counter(); ok = c.matches(counts); t2 = time.perf_counter() - t
    pool = np.arange(g, dtype=np.int32)
    t = time.perf_counter(); c.corr(pool); t3 = time.perf_counter() - t
    t = time.perf_counter(); f = _hostpar.col_stats_first(counts); t4 = time.perf_counter() - t
    t = time.perf_counter(); v = _hostpar.col_stats_var(counts, f["avg"]); t5 = time.perf_counter() - t
    print("create %.3f  matches %.3f (%s)  corr %.3f  stats first %.3f  var %.3f" % (t1, t2, ok, t3, t4, t5), flush=True)
    c.close()
