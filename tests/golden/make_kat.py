"""Generate tests/golden/kat_steps.npz: known-answer vectors for the MultiNet hot path.

Run HERE (survey/build container, CPU): `python tests/golden/make_kat.py`.

Independent restatement used to PIN the CPU oracle: torch.float64 tensors, torch.autograd
for every gradient, torch's own relu/softplus, and a hand-written Keras-form Adam
(TF ResourceApplyAdam: eps outside the bias correction -- torch.optim.Adam is NOT used, its
eps placement differs).  The semantics restated are those of the
reference's Keras calls (deepimpute/multinet.py:36-41 wMSE, :126-148 topology, :164 Adam,
:238-244 fit, :278 predict).  TensorFlow itself is not installable in this image, so this is
the strongest pin available; the npz holds only inputs and expected outputs.
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_steps.npz")


def main():
    rng = np.random.default_rng(20240928)
    n, g = 96, 220
    H, O, B = 32, 48, 64
    Ds = [97, 64, 33]
    K = len(Ds)
    p = 0.2
    lr, b1c, b2c, eps = 1e-3, 0.9, 0.999, 1e-7

    # log1p(count)-like data with ~45% zeros, float32-exact
    lam = rng.lognormal(0.5, 1.2, size=g)
    counts = rng.poisson(rng.gamma(2.0, lam / 2.0, size=(n, g)))
    norm = np.log1p(counts).astype(np.float32)

    pred = [rng.choice(g, D, replace=False).astype(np.int32) for D in Ds]
    targ = [rng.choice(g, O, replace=False).astype(np.int32) for _ in Ds]
    W1 = [(rng.uniform(-1, 1, (D, H)) * np.sqrt(6.0 / (D + H))).astype(np.float32) for D in Ds]
    b1 = [(0.05 * rng.standard_normal(H)).astype(np.float32) for _ in Ds]
    W2 = [(rng.uniform(-1, 1, (H, O)) * np.sqrt(6.0 / (H + O))).astype(np.float32) for _ in Ds]
    b2 = [(0.05 * rng.standard_normal(O)).astype(np.float32) for _ in Ds]

    # three optimiser steps: two full batches and a partial one (37 rows)
    batches = [rng.choice(n, B, replace=False).astype(np.int32),
               rng.choice(n, B, replace=False).astype(np.int32),
               rng.choice(n, 37, replace=False).astype(np.int32)]
    masks = [(rng.random((K, len(r), H)) >= p).astype(np.uint8) for r in batches]
    val_rows = rng.choice(n, 20, replace=False).astype(np.int32)

    T = torch.float64
    normt = torch.tensor(norm, dtype=T)
    out = dict(norm=norm, H=H, O=O, B=B, p=np.float32(p), lr=np.float32(lr), beta1=np.float32(b1c),
               beta2=np.float32(b2c), eps=np.float32(eps), Ds=np.array(Ds, np.int32), val_rows=val_rows)
    # use the float32-rounded hyper-parameters exactly as the C side sees them
    lr_, b1_, b2_, eps_ = float(np.float32(lr)), float(np.float32(b1c)), float(np.float32(b2c)), float(np.float32(eps))
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))

    for k in range(K):
        params = [torch.tensor(a, dtype=T, requires_grad=True) for a in (W1[k], b1[k], W2[k], b2[k])]
        m = [torch.zeros_like(q) for q in params]
        v = [torch.zeros_like(q) for q in params]
        losses, grads0 = [], None
        for t, (rows, mask) in enumerate(zip(batches, masks), start=1):
            x = normt[torch.tensor(rows.astype(np.int64))][:, torch.tensor(pred[k].astype(np.int64))]
            y = normt[torch.tensor(rows.astype(np.int64))][:, torch.tensor(targ[k].astype(np.int64))]
            keep = torch.tensor(mask[k], dtype=T)
            a = x @ params[0] + params[1]
            dd = torch.relu(a) * keep * scale
            z = dd @ params[2] + params[3]
            yhat = torch.nn.functional.softplus(z)
            loss = torch.mean(y * (y - yhat) ** 2)            # wMSE, weights = y_true
            gs = torch.autograd.grad(loss, params)
            if t == 1:
                grads0 = [gq.detach().numpy().copy() for gq in gs]
            losses.append(float(loss.detach()))
            alpha = lr_ * np.sqrt(1.0 - b2_ ** t) / (1.0 - b1_ ** t)
            alpha = float(np.float32(alpha))                   # the C side passes alpha as fp32
            with torch.no_grad():
                for q, gq, mq, vq in zip(params, gs, m, v):
                    mq += (gq - mq) * (1.0 - b1_)
                    vq += (gq * gq - vq) * (1.0 - b2_)
                    q -= (mq * alpha) / (vq.sqrt() + eps_)
        with torch.no_grad():
            xa = normt[:, torch.tensor(pred[k].astype(np.int64))]
            za = torch.relu(xa @ params[0] + params[1]) @ params[2] + params[3]
            pred_all = torch.nn.functional.softplus(za).numpy()
            yv = normt[torch.tensor(val_rows.astype(np.int64))][:, torch.tensor(targ[k].astype(np.int64))].numpy()
            pv = pred_all[val_rows]
            vloss = float(np.mean(yv * (yv - pv) ** 2))
        out.update({
            "pred%d" % k: pred[k], "targ%d" % k: targ[k],
            "W1_%d" % k: W1[k], "b1_%d" % k: b1[k], "W2_%d" % k: W2[k], "b2_%d" % k: b2[k],
            "loss_%d" % k: np.array(losses),
            "val_loss_%d" % k: np.float64(vloss),
            "pred_all_%d" % k: pred_all,
        })
        for name, arr in zip(("W1", "b1", "W2", "b2"), grads0):
            out["g0_%s_%d" % (name, k)] = arr
        for name, q, mq, vq in zip(("W1", "b1", "W2", "b2"), params, m, v):
            out["out_%s_%d" % (name, k)] = q.detach().numpy()
            out["m_%s_%d" % (name, k)] = mq.numpy()
            out["v_%s_%d" % (name, k)] = vq.numpy()
    for t, (rows, mask) in enumerate(zip(batches, masks)):
        out["rows_%d" % t] = rows
        out["mask_%d" % t] = mask
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
