"""Generate tests/golden/kat_act.npz: known answers for the hidden activations other than relu.

Run HERE (CPU): `python tests/golden/make_act.py`.  Same independent restatement as make_kat.py (torch.float64,
torch.autograd for every gradient, torch's own activation functions, hand-written Keras-form Adam), one tiny
sub-net per activation, two optimiser steps with injected batches and dropout masks (the second one partial).
The reference builds the hidden layer as Dense(neurons, activation=layer['activation'])
(deepimpute/multinet.py:137); Keras' elu uses alpha = 1.  The npz holds inputs and expected outputs only.
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_act.npz")
ACTS = {"linear": lambda a: a, "sigmoid": torch.sigmoid, "tanh": torch.tanh,
        "elu": torch.nn.functional.elu, "softplus": torch.nn.functional.softplus,
        # round 5: the rest of keras.activations' element-wise names, in their TF / Keras 2.x definitions
        "selu": torch.nn.functional.selu, "softsign": torch.nn.functional.softsign, "swish": torch.nn.functional.silu,
        "gelu": lambda a: torch.nn.functional.gelu(a, approximate="none"), "exponential": torch.exp,
        "hard_sigmoid": lambda a: torch.clamp(0.2 * a + 0.5, 0.0, 1.0)}


def main():
    rng = np.random.default_rng(314)
    n, g, D, H, O, B = 80, 120, 37, 20, 16, 32
    p, lr, b1c, b2c, eps = 0.3, 2e-3, 0.9, 0.999, 1e-7
    lam = rng.lognormal(0.5, 1.2, size=g)
    norm = np.log1p(rng.poisson(rng.gamma(2.0, lam / 2.0, size=(n, g)))).astype(np.float32)
    pred = rng.choice(g, D, replace=False).astype(np.int32)
    targ = rng.choice(g, O, replace=False).astype(np.int32)
    init = [(rng.uniform(-1, 1, (D, H)) * np.sqrt(6.0 / (D + H))).astype(np.float32), (0.1 * rng.standard_normal(H)).astype(np.float32),
            (rng.uniform(-1, 1, (H, O)) * np.sqrt(6.0 / (H + O))).astype(np.float32), (0.1 * rng.standard_normal(O)).astype(np.float32)]
    batches = [rng.choice(n, B, replace=False).astype(np.int32), rng.choice(n, 19, replace=False).astype(np.int32)]
    masks = [(rng.random((1, len(r), H)) >= p).astype(np.uint8) for r in batches]
    T = torch.float64
    lr_, b1_, b2_, eps_ = (float(np.float32(x)) for x in (lr, b1c, b2c, eps))
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    X = torch.tensor(norm, dtype=T)[:, torch.tensor(pred.astype(np.int64))]
    Y = torch.tensor(norm, dtype=T)[:, torch.tensor(targ.astype(np.int64))]
    out = dict(norm=norm, pred=pred, targ=targ, D=np.int32(D), H=np.int32(H), O=np.int32(O), B=np.int32(B), p=np.float32(p),
               lr=np.float32(lr), beta1=np.float32(b1c), beta2=np.float32(b2c), eps=np.float32(eps),
               rows_0=batches[0], rows_1=batches[1], mask_0=masks[0], mask_1=masks[1], acts=np.array(sorted(ACTS)))
    for nm, a in zip(("W1", "b1", "W2", "b2"), init):
        out["init_" + nm] = a
    for name, fn in ACTS.items():
        params = [torch.tensor(a, dtype=T, requires_grad=True) for a in init]
        m = [torch.zeros_like(q) for q in params]
        v = [torch.zeros_like(q) for q in params]
        losses = []
        for t, (rows, mask) in enumerate(zip(batches, masks), start=1):
            r = torch.tensor(rows.astype(np.int64))
            hdn = fn(X[r] @ params[0] + params[1]) * torch.tensor(mask[0], dtype=T) * scale
            yh, y = torch.nn.functional.softplus(hdn @ params[2] + params[3]), Y[r]
            loss = (y * (y - yh) ** 2).mean()
            for q in params:
                q.grad = None
            loss.backward()
            alpha = lr_ * np.sqrt(1.0 - b2_ ** t) / (1.0 - b1_ ** t)
            with torch.no_grad():
                for q, mq, vq in zip(params, m, v):
                    mq += (q.grad - mq) * (1.0 - b1_)
                    vq += (q.grad * q.grad - vq) * (1.0 - b2_)
                    q -= alpha * mq / (vq.sqrt() + eps_)
            losses.append(float(loss.detach()))
        out[name + "/loss"] = np.array(losses)
        with torch.no_grad():
            out[name + "/predict"] = torch.nn.functional.softplus(fn(X @ params[0] + params[1]) @ params[2] + params[3]).numpy()
        for nm, q in zip(("W1", "b1", "W2", "b2"), params):
            out["%s/%s" % (name, nm)] = q.detach().numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
