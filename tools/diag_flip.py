"""Diagnostic: where do the resident kernel and the oracle part ways on configs[1]?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "32")
import numpy as np
import bench
from helpers import relu_flip_units
from deepimpute_amd.engine import HipEngine
from oracle.dimo import OracleEngine

cfg = bench.CONFIGS["cfg2"]
norm = bench.synth_counts(cfg["n"], cfg["g"], seed=0)
targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=0)
train, val = bench.split_rows(cfg["n"], seed=0)
K = 10
kw = dict(batch_size=64, dropout_rate=0.2, learning_rate=1e-3, seed=1234)

def build(cls, ntr, ks=range(K), **extra):
    ks = list(ks)
    e = cls([len(preds[k]) for k in ks], cfg["H"], cfg["O"], subnet_offset=ks[0], **kw, **extra)
    e.set_matrix(norm)
    for i, k in enumerate(ks):
        e.set_indices(i, preds[k], targets[k])
    e.gather(True); e.set_split(train[:ntr], val); e.init_weights()
    return e

for res in ("1", "0"):
    os.environ["DIMN_RESIDENT"] = res
    for ntr in (64, 128, 192, 213):
        a, b = build(HipEngine, ntr), build(OracleEngine, ntr)
        a.train_epoch(0); b.train_epoch(0)
        bad = {k: relu_flip_units(a, b, k).tolist() for k in range(K)}
        bad = {k: v for k, v in bad.items() if v}
        print("resident", res, "train rows", ntr, "bad units", bad, flush=True)
        if bad and ntr == 213 and res == "1":
            k = 3
            Wa, ba = a.get_weights(k)[:2]; Wb, bb = b.get_weights(k)[:2]
            for u in bad.get(k, []):
                d = np.abs(Wa[:, u] - Wb[:, u])
                print(" unit", u, "max|dW1|", d.max(), "n(|dW1|>2e-5)", int((d > 2e-5).sum()), "db1", ba[u] - bb[u], "W1 norm", np.abs(Wb[:, u]).max())
        a.close(); b.close()

# pre-activations of the flagged units of sub-net 3 in every batch, fp64 replay
o = build(OracleEngine, 213, ks=[3], fp64=True)
perm = o.epoch_permutation(0)
tr = train[:213]
for t in range(4):
    rows = tr[perm[t * 64:(t + 1) * 64]]
    W1, b1 = o.get_weights(0)[:2]
    X = norm[rows][:, preds[3]].astype(np.float64)
    for u in (52, 120, 180):
        w = W1[:, u].astype(np.float64)
        a_ = X @ w + float(b1[u])
        bound = np.finfo(np.float32).eps * (np.abs(X) @ np.abs(w) + abs(float(b1[u])))
        i = int(np.argmin(np.abs(a_) / bound))
        print("step", t, "unit", u, "min |a|/eps-bound", abs(a_[i]) / bound[i], "a", a_[i], "n(a>0)", int((a_ > 0).sum()), "of", len(rows), flush=True)
    o.train_step(rows, epoch_key=0, step_key=t, want_loss=False)
