#!/usr/bin/env python
"""Run the drop-in MultiNet end to end on a synthetic counts matrix (BASELINE configs[1]: 5k x 5k by
default) and print where the time goes: host planning (reference-identical numpy/pandas code), GPU
fit, predict + host post-processing."""
import argparse
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepimpute_amd import multinet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=5000)
    ap.add_argument("--genes", type=int, default=5000)
    ap.add_argument("--max-epochs", type=int, default=500)
    args = ap.parse_args()
    counts = np.expm1(bench.synth_counts(args.cells, args.genes, seed=0)).round()
    raw = pd.DataFrame(counts, index=["c%d" % i for i in range(args.cells)], columns=["g%d" % j for j in range(args.genes)])
    marks = {}
    for name in ("get_distance_matrix",):
        fn = getattr(multinet, name)
        def timed(*a, _fn=fn, _name=name, **k):
            t = time.time(); out = _fn(*a, **k); marks[_name] = time.time() - t; return out
        setattr(multinet, name, timed)
    net = multinet.MultiNet(verbose=0, max_epochs=args.max_epochs)
    orig_sp = net.setPredictors
    def sp(*a, **k):
        t = time.time(); out = orig_sp(*a, **k); marks["setPredictors"] = time.time() - t; return out
    net.setPredictors = sp
    t0 = time.time()
    net.fit(raw, NN_lim=args.genes)
    t_fit = time.time() - t0
    t0 = time.time()
    out = net.predict(raw)
    t_pred = time.time() - t0
    print("cells=%d genes=%d K=%d epochs=%d" % (args.cells, args.genes, len(net.predictors), net.trained_epochs))
    print("fit total %.2fs (corr matrix %.2fs, setPredictors %.2fs)  predict total %.2fs" %
          (t_fit, marks.get("get_distance_matrix", 0), marks.get("setPredictors", 0), t_pred))
    print("test_metrics", net.test_metrics, "val_loss first/last %.4f %.4f" % (net.history["val_loss"][0], net.history["val_loss"][-1]))
    print("cells/s end to end (drop-in, incl. host planning): %.1f" % (args.cells / (t_fit + t_pred)))
    assert out.shape == raw.shape


if __name__ == "__main__":
    main()
