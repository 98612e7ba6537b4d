"""Where the planning seconds of the drop-in fit go (50k x 20k counts): each stage ALONE, then the pairs that run concurrently in fit().
    python tools/dropin_probe.py"""
import os, sys, time, threading, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from deepimpute_amd._counts import DeviceCounts
from deepimpute_amd import _hostpar

print(subprocess.run("lscpu | grep -i 'model name\\|socket\\|numa\\|^CPU(s)'; free -g | head -2", shell=True, capture_output=True, text=True).stdout, flush=True)
n, g = 50000, 20000
counts = np.rint(np.expm1(bench.synth_counts(n, g, seed=0).astype(np.float64)))
print("max count", counts.max(), flush=True)
pool = np.arange(g, dtype=np.int32)


def clock(fn):
    t = time.perf_counter(); r = fn(); return time.perf_counter() - t, r


for rep in range(2):
    t1, c = clock(lambda: DeviceCounts.try_create(counts, 0))
    t3, _ = clock(lambda: c.corr(pool))
    t4, f = clock(lambda: _hostpar.col_stats_first(counts))
    t5, v = clock(lambda: _hostpar.col_stats_var(counts, f["avg"]))
    t2, ok = clock(lambda: c.matches(counts))
    t6, st = clock(lambda: c.gene_stats())
    os.environ["DIMN_CORR_I8"] = "0"
    t7, _ = clock(lambda: c.corr(pool))
    os.environ.pop("DIMN_CORR_I8")
    print("alone: create %.3f  corr (int8) %.3f  corr (float64) %.3f  host stats first %.3f  var %.3f  checksum %.3f  device stats %.3f (mean equal %s, var equal %s)"
          % (t1, t3, t7, t4, t5, t2, t6, np.array_equal(st["mean"], f["mean"]), np.array_equal(st["var"], v)), flush=True)
    c.close()
for threads in (8, 16, 32, 64, 128):
    os.environ["DIMN_COUNTS_THREADS_REMOVED"] = str(threads)
    t1, c = clock(lambda: DeviceCounts.try_create(counts, 0))
    t2, ok = clock(lambda: c.matches(counts))
    print("DIMN_COUNTS_THREADS_REMOVED=%d: create %.3f checksum-only scan %.3f" % (threads, t1, t2), flush=True)
    c.close()
os.environ.pop("DIMN_COUNTS_THREADS_REMOVED")
# concurrent: create || first
box = {}
th = threading.Thread(target=lambda: box.__setitem__("c", clock(lambda: DeviceCounts.try_create(counts, 0))))
t = time.perf_counter(); th.start(); t4, f = clock(lambda: _hostpar.col_stats_first(counts)); th.join(); tot = time.perf_counter() - t
print("concurrent: create %.3f || stats first %.3f -> %.3f" % (box["c"][0], t4, tot), flush=True)
c = box["c"][1]
th = threading.Thread(target=lambda: box.__setitem__("k", clock(lambda: c.corr(pool))))
t = time.perf_counter(); th.start(); t5, v = clock(lambda: _hostpar.col_stats_var(counts, f["avg"])); th.join(); tot = time.perf_counter() - t
print("concurrent: corr %.3f || stats var %.3f -> %.3f" % (box["k"][0], t5, tot), flush=True)
c.close()
