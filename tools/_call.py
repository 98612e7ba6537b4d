import os, sys, time
import numpy as np, pandas as pd
sys.path.insert(0, ".")
import bench
from deepimpute_amd import multinet
n, g = 50000, 20000
counts = np.rint(np.expm1(bench.synth_counts(n, g, seed=0).astype(np.float64))).astype(np.int64)
raw = pd.DataFrame(counts, index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
H = int(os.environ.get("HH", "300"))
net = multinet.MultiNet(verbose=0, learning_rate=5e-4, max_epochs=int(os.environ.get("ME", "300")), architecture=[{"type": "dense", "activation": "relu", "neurons": H}, {"type": "dropout", "activation": "dropout", "rate": 0.2}])
t0 = time.time(); net.fit(raw, NN_lim=g); t = time.time() - t0
h = net.history
print("H", H, "env", {k: v for k, v in os.environ.items() if k.startswith("DIMN_")}, "epochs", net.trained_epochs, "fit %.2f s" % t)
print("val_loss", " ".join("%.5f" % v for v in h["val_loss"]))
print("loss    ", " ".join("%.5f" % v for v in h["loss"]))
print("test_metrics", net.test_metrics)
