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
    counts = np.rint(np.expm1(bench.synth_counts(args.cells, args.genes, seed=0).astype(np.float64)))      # float64, as pd.read_csv gives
    raw = pd.DataFrame(counts, index=["c%d" % i for i in range(args.cells)], columns=["g%d" % j for j in range(args.genes)])
    marks = {}

    def clock(owner, name, label=None):
        fn = getattr(owner, name)
        def timed(*a, **k):
            t = time.time(); out = fn(*a, **k); marks[label or name] = marks.get(label or name, 0.0) + time.time() - t; return out
        setattr(owner, name, timed)

    from deepimpute_amd import _hostpar, engine as eng_mod
    clock(multinet, "get_distance_matrix"); clock(multinet, "_abs_corrcoef"); clock(multinet, "inspect_data")
    for name in ("column_var_mean", "log1p_float32", "zero_nans_inplace", "take_columns"):
        clock(_hostpar, name)
    for name in ("set_matrix", "gather", "fit", "predict", "get_weights", "init_weights", "impute_finish", "val_metrics", "predict_device"):
        clock(eng_mod.HipEngine, name, "engine." + name)
    net = multinet.MultiNet(verbose=0, max_epochs=args.max_epochs)
    clock(net, "setPredictors"); clock(net, "save"); clock(net, "_held_out_metrics"); clock(net, "_set_predictors_device")
    clock(net, "filter_genes"); clock(net, "setTargets"); clock(net, "_bind_columns"); clock(multinet, "_candidate_pool")
    t0 = time.time()
    net.fit(raw, NN_lim=args.genes)
    t_fit = time.time() - t0
    t0 = time.time()
    out = net.predict(raw)
    t_pred = time.time() - t0
    print("cells=%d genes=%d K=%d epochs=%d" % (args.cells, args.genes, len(net.predictors), net.trained_epochs))
    print("fit total %.2fs  predict total %.2fs" % (t_fit, t_pred))
    print("  " + "  ".join("%s %.2f" % kv for kv in sorted(marks.items(), key=lambda kv: -kv[1])))
    print("test_metrics", net.test_metrics, "val_loss first/last %.4f %.4f" % (net.history["val_loss"][0], net.history["val_loss"][-1]))
    print("cells/s end to end (drop-in, incl. host planning): %.1f" % (args.cells / (t_fit + t_pred)))
    assert out.shape == raw.shape


if __name__ == "__main__":
    main()
