#!/usr/bin/env python
"""cProfile of the drop-in MultiNet.fit + predict at 50k x 20k (2 epochs): where the HOST time goes."""
import cProfile
import io
import os
import pstats
import sys
import contextlib

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepimpute_amd.multinet import MultiNet  # noqa: E402

n, g = int(sys.argv[1]) if len(sys.argv) > 1 else 50000, int(sys.argv[2]) if len(sys.argv) > 2 else 20000
counts = np.rint(np.expm1(bench.synth_counts(n, g, seed=0).astype(np.float64)))
raw = pd.DataFrame(counts, index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
net = MultiNet(verbose=0, max_epochs=2)
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    net.fit(raw, NN_lim=g)
    out = net.predict(raw)
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
