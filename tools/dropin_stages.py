#!/usr/bin/env python
"""Wall time of every stage of the drop-in MultiNet.fit + predict at 50k x 20k (E epochs): which host / PCIe work surrounds the engine."""
import contextlib
import io
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepimpute_amd import _hostpar, multinet  # noqa: E402
from deepimpute_amd.multinet import MultiNet  # noqa: E402

n, g = int(sys.argv[1]) if len(sys.argv) > 1 else 50000, int(sys.argv[2]) if len(sys.argv) > 2 else 20000
E = int(sys.argv[3]) if len(sys.argv) > 3 else 2
counts = np.rint(np.expm1(bench.synth_counts(n, g, seed=0).astype(np.float64)))
raw = pd.DataFrame(counts, index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
log = []


def wrap(obj, name, label=None):
    fn = getattr(obj, name)

    def inner(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            log.append((label or name, time.perf_counter() - t0))
    setattr(obj, name, inner)


for name in dir(_hostpar):
    if not name.startswith("_") and callable(getattr(_hostpar, name)) and getattr(getattr(_hostpar, name), "__module__", "") == _hostpar.__name__:
        wrap(_hostpar, name, "_hostpar." + name)
for name in ("inspect_data", "get_distance_matrix"):
    if hasattr(multinet, name):
        wrap(multinet, name)
for name in ("filter_genes", "setTargets", "setPredictors", "_set_predictors_device", "_build_shard", "_bind_columns", "_hand_over", "save", "_held_out_metrics",
             "load", "_finish_on_device", "_finish_on_host", "_release_engine"):
    if hasattr(MultiNet, name):
        wrap(MultiNet, name)
from deepimpute_amd.engine import HipEngine  # noqa: E402
for name in ("fit", "init_weights", "set_split", "impute_finish", "predict_device", "set_matrix", "gather"):
    wrap(HipEngine, name, "engine." + name)
for rep in range(2):
    del log[:]
    net = MultiNet(verbose=0, max_epochs=E, patience=10 ** 6)
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter(); net.fit(raw, NN_lim=g); t1 = time.perf_counter(); out = net.predict(raw); t2 = time.perf_counter()
    print("run %d: fit %.3f s, predict %.3f s, total %.3f s -> %.0f cells/s" % (rep, t1 - t0, t2 - t1, t2 - t0, n / (t2 - t0)))
    for label, dt in net.timings.items():
        print("   %-36s %7.3f" % (label, dt))
    print("   (sum of stages %.3f)" % sum(net.timings.values()))
    net.close()
