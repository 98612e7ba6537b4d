"""Random shapes through the general path twice -- k_gen_rowgemm (default) and the LDS-staged k_gen_gemm (DIMN_RES_TEST=gemm=0) -- and the
difference of what they train (one epoch + validation + prediction).  Widths that are and are not multiples of 4 / 16 / 64, ragged predictor
counts (the masked last chunk of K), batches of 1 .. 200 (row clamps, several 64-row blocks), one to three hidden layers, every loss.
    python tools/gen_shape_sweep.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from helpers import make_problem          # noqa: E402
from deepimpute_amd.engine import HipGeneralEngine      # noqa: E402

ACTS = ["relu", "tanh", "sigmoid", "elu", "gelu", "linear"]
LOSSES = ["wmse", "mse", "mae", "huber", "wmse_binary"]


def run(prob, layers, env, **kw):
    if env:
        os.environ["DIMN_RES_TEST"] = env
    else:
        os.environ.pop("DIMN_RES_TEST", None)
    e = HipGeneralEngine(prob["Ds"], layers, prob["O"], **kw)
    e.set_matrix(prob["norm"])
    for k in range(len(prob["Ds"])):
        e.set_indices(k, prob["pred"][k], prob["targ"][k])
    e.gather(True)
    e.set_split(prob["train"], prob["val"])
    e.init_weights()
    tl = np.asarray(e.train_epoch(0))
    vl = np.asarray(e.val_loss())
    pr = e.predict()
    ws = [w for k in range(e.K) for w in e.get_weights(k)]
    e.close()
    return tl, vl, pr, ws


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst = 0.0
    for c in range(cases):
        K = int(rng.integers(1, 10))
        Ds = [int(x) for x in rng.integers(5, 700, size=K)]
        O = int(rng.choice([4, 12, 64, 100, 257, 512]))
        nl = int(rng.integers(1, 4))
        layers = [(int(rng.choice([4, 8, 20, 50, 64, 100, 132, 256, 300])), str(rng.choice(ACTS)), float(rng.choice([0.0, 0.2]))) for _ in range(nl)]
        B = int(rng.choice([1, 7, 33, 64, 65, 128, 200]))
        n = B * 2 + int(rng.integers(1, B + 2)) + 40
        prob = make_problem(n=n, g=800, Ds=Ds, H=layers[0][0], O=O, seed=int(rng.integers(1 << 30)))
        kw = dict(batch_size=B, learning_rate=1e-3, seed=int(rng.integers(1 << 30)), loss=str(rng.choice(LOSSES)))
        a = run(prob, layers, None, **kw)
        b = run(prob, layers, "gemm=0", **kw)
        d_loss = float(np.max(np.abs(a[0] - b[0]) / np.maximum(np.abs(b[0]), 1e-12)))
        d_val = float(np.max(np.abs(a[1] - b[1]) / np.maximum(np.abs(b[1]), 1e-12)))
        d_pred = float(np.max(np.abs(a[2] - b[2]) / (np.abs(b[2]) + 1e-3)))
        d_w = max(float(np.max(np.abs(x - y))) for x, y in zip(a[3], b[3]))
        ok = d_loss < 1e-5 and d_val < 1e-5 and d_pred < 1e-3 and d_w < 5e-5 and all(np.isfinite(x).all() for x in a[3])
        worst = max(worst, d_loss, d_val)
        print("%2d K=%d D=%s O=%d layers=%s B=%d n=%d %s: loss %.1e val %.1e pred %.1e w %.1e %s" % (c, K, Ds[:3], O, layers, B, n, kw["loss"], d_loss, d_val, d_pred, d_w, "ok" if ok else "MISMATCH"))
        if not ok:
            sys.exit(1)
    print("all %d cases agree; worst relative loss difference %.1e" % (cases, worst))


if __name__ == "__main__":
    main()
