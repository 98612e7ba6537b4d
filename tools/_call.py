import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from deepimpute_amd.engine import HipEngine
from oracle.dimo import OracleEngine
H = int(os.environ.get("HH", "272"))
cfg = dict(bench.CONFIGS["cfg3"]); cfg["H"] = H
norm = bench.synth_counts(2048, cfg["g"], seed=0)
targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=0)
K = targets.shape[0]
train = np.arange(0, 3 * 64 + 21, dtype=np.int32) * 7 % 1700
val = np.arange(1700, 1950, dtype=np.int32)
kw = dict(batch_size=64, dropout_rate=0.2, learning_rate=1e-3, seed=1234)
def load(cls, ks, **k2):
    e = cls([len(preds[k]) for k in ks], H, cfg["O"], subnet_offset=ks[0], **kw, **k2)
    for i, k in enumerate(ks): e.set_indices(i, preds[k], targets[k])
    e.set_matrix(norm); e.gather(True); e.set_split(train, val); e.init_weights(); return e
a = load(HipEngine, list(range(K)))
print(a.path_info())
b = load(OracleEngine, [37]); b64 = load(OracleEngine, [37], fp64=True)
nsteps = int(os.environ.get("NSTEPS", "4"))
B = 64
for e_ in (a, b, b64): pass
la = a.train_epoch(0); lb = b.train_epoch(0); l64 = b64.train_epoch(0)
print("loss", la[37], lb[0], l64[0])
wa = a.get_weights(37); wb = b.get_weights(0); w64 = b64.get_weights(0)
for name, x, y, z in zip(("W1", "b1", "W2", "b2"), wa, wb, w64):
    d = np.abs(x - y); d64 = np.abs(x - z); o64 = np.abs(y - z)
    print(name, "hip-vs-o32 max %.3e  hip-vs-o64 max %.3e  o32-vs-o64 max %.3e" % (d.max(), d64.max(), o64.max()))
W1a, W1b, W1c = wa[0], wb[0], w64[0]
col = np.abs(W1a - W1b).max(axis=0)
bad = np.argsort(col)[-5:]
print("worst units", bad, col[bad])
u = int(bad[-1])
d = W1a[:, u] - W1b[:, u]
print("unit", u, "rows off (>2e-5):", int((np.abs(d) > 2e-5).sum()), "of", d.size, "max", np.abs(d).max(), "hip-vs-64", np.abs(W1a[:, u] - W1c[:, u]).max(), "o32-vs-64", np.abs(W1b[:, u] - W1c[:, u]).max())
idx = np.argsort(np.abs(d))[-8:]
print("largest diffs at d =", idx, d[idx], "hip", W1a[idx, u], "o32", W1b[idx, u], "o64", W1c[idx, u])
print("b1 unit", wa[1][u], wb[1][u], w64[1][u])
