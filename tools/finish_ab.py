"""Same-box A/B of predict()'s epilogue at the bench size: dense (8 GB device to host) vs restore (only the zero entries), host threads."""
import contextlib, io, os, sys, time
import numpy as np
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deepimpute_amd.multinet import MultiNet

n, g = 50000, 20000
norm = bench.synth_counts(n, g, seed=0)
raw = pd.DataFrame(np.rint(np.expm1(norm.astype(np.float64))), index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
with contextlib.redirect_stdout(io.StringIO()):
    net = MultiNet(verbose=0, max_epochs=1, patience=10 ** 6)
    net.fit(raw, NN_lim=g)
    ref = None
    for rep in range(4):
        for label, restore in (("dense", False), ("restore", True)):
            type(net._engine).restore_epilogue = restore
            t0 = time.perf_counter()
            out = net.predict(raw)
            dt = time.perf_counter() - t0
            if ref is None:
                ref = out.values.copy()
            same = np.array_equal(out.values, ref)
            sys.stderr.write("rep %d  %-22s predict %.3f s  forward+finish %.3f s  same as dense: %s\n" % (rep, label, dt, net.timings["predict.forward+finish"], same))
            del out
net.close()
