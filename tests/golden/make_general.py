"""Generate tests/golden/kat_general.npz: known answers for the GENERAL path (any architecture / batch size / loss).

Run HERE (CPU): `python tests/golden/make_general.py`.

Pins oracle/dimo_general.c independently of it: torch.float64 autograd for every gradient, torch's own tanh / relu /
softplus, a hand-written Keras-form Adam (eps outside the bias correction), the four losses as Keras defines them
(reference deepimpute/multinet.py:36-41 wMSE; keras.losses.mean_squared_error / mean_absolute_error: mean over the last
axis, then over the batch), and the Philox dropout streams restated in numpy (make_epochs.py, checked there against
the Random123 vectors) with the dropout-layer ordinal in the top byte of the step word.
Problem: K = 2 ragged sub-nets, architecture Dense(24, tanh) - Dropout(0.1) - Dense(16, relu) - Dropout(0.25) -
Dense(12, softplus), batch size 100 (> 64): two optimiser steps (100 rows, then a partial batch of 37), then predict.
"""
import os

import numpy as np
import torch

from make_epochs import dropout_keep

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_general.npz")
LOSSES = ("wmse", "wmse_binary", "mse", "mae", "msle", "logcosh", "huber", "poisson")


def loss_fn(name, y, yh):
    if name == "mae":
        return torch.mean(torch.abs(y - yh))
    if name == "msle":                                           # keras.losses.mean_squared_logarithmic_error (epsilon 1e-7)
        return torch.mean((torch.log(torch.clamp(y, min=1e-7) + 1.0) - torch.log(torch.clamp(yh, min=1e-7) + 1.0)) ** 2)
    if name == "logcosh":                                        # keras.losses.logcosh: x + softplus(-2x) - log 2
        x = yh - y
        return torch.mean(x + torch.nn.functional.softplus(-2.0 * x) - np.log(2.0))
    if name == "huber":                                          # keras.losses.huber, delta = 1
        e = yh - y
        return torch.mean(torch.where(e.abs() <= 1.0, 0.5 * e * e, e.abs() - 0.5))
    if name == "poisson":                                        # keras.losses.poisson
        return torch.mean(yh - y * torch.log(yh + 1e-7))
    w = y if name == "wmse" else ((y > 0).to(y.dtype) if name == "wmse_binary" else torch.ones_like(y))
    return torch.mean(w * (y - yh) ** 2)


def main():
    rng = np.random.default_rng(4242)
    n, g, O, B = 160, 140, 12, 100
    Ds, widths, acts, rates = [31, 18], [24, 16], ["tanh", "relu"], [0.1, 0.25]
    K, seed, kg0 = len(Ds), 99, 3
    lr, b1c, b2c, eps = 2e-3, 0.9, 0.999, 1e-7
    lam = rng.lognormal(0.5, 1.2, size=g)
    norm = np.log1p(rng.poisson(rng.gamma(2.0, lam / 2.0, size=(n, g)))).astype(np.float32)
    pred = [rng.choice(g, D, replace=False).astype(np.int32) for D in Ds]
    targ = [rng.choice(g, O, replace=False).astype(np.int32) for _ in Ds]
    batches = [rng.choice(n, 100, replace=False).astype(np.int32), rng.choice(n, 37, replace=False).astype(np.int32)]
    out = dict(norm=norm, Ds=np.array(Ds, np.int32), widths=np.array(widths, np.int32), acts=np.array(acts), rates=np.array(rates, np.float32),
               O=O, B=B, seed=seed, subnet_offset=kg0, lr=np.float32(lr), beta1=np.float32(b1c), beta2=np.float32(b2c), eps=np.float32(eps),
               rows_0=batches[0], rows_1=batches[1], losses=np.array(LOSSES))
    dims = lambda D: [D] + widths + [O]
    init = {}
    for k in range(K):
        d = dims(Ds[k])
        for l in range(3):
            init[(k, l)] = ((rng.uniform(-1, 1, (d[l], d[l + 1])) * np.sqrt(6.0 / (d[l] + d[l + 1]))).astype(np.float32),
                            (0.05 * rng.standard_normal(d[l + 1])).astype(np.float32))
            out["pred%d" % k], out["targ%d" % k] = pred[k], targ[k]
            out["init_W_%d_%d" % (k, l)], out["init_b_%d_%d" % (k, l)] = init[(k, l)]
    T = torch.float64
    lr_, b1_, b2_, eps_ = (float(np.float32(x)) for x in (lr, b1c, b2c, eps))
    normt = torch.tensor(norm, dtype=T)
    fa = {"tanh": torch.tanh, "relu": torch.relu}
    in_rate = 0.15                       # "wmse+input_dropout": the same model with a Dropout(0.15) layer BEFORE the first Dense layer
    out["input_dropout_rate"] = np.float32(in_rate)
    for name in LOSSES + ("wmse+input_dropout",):
        indrop = name.endswith("+input_dropout")
        for k in range(K):
            params = []
            for l in range(3):
                params += [torch.tensor(init[(k, l)][0], dtype=T, requires_grad=True), torch.tensor(init[(k, l)][1], dtype=T, requires_grad=True)]
            m = [torch.zeros_like(p) for p in params]
            v = [torch.zeros_like(p) for p in params]

            def forward(x, masks, in_mask=None):
                h = x
                if in_mask is not None:
                    h = h * torch.tensor(in_mask, dtype=T) * float(np.float32(1.0) / (np.float32(1.0) - np.float32(in_rate)))
                for l in range(2):
                    h = fa[acts[l]](h @ params[2 * l] + params[2 * l + 1])
                    if masks is not None:
                        scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(rates[l])))
                        h = h * torch.tensor(masks[l], dtype=T) * scale
                return torch.nn.functional.softplus(h @ params[4] + params[5])

            losses = []
            for t, rows in enumerate(batches):
                first = 1 if indrop else 0          # dropout ordinals count the Dropout layers of the architecture in order
                masks = [dropout_keep(seed, kg0 + k, 0, t | ((dl + first) << 24), len(rows), widths[dl], rates[dl]) for dl in range(2)]
                in_mask = dropout_keep(seed, kg0 + k, 0, t, len(rows), Ds[k], in_rate) if indrop else None
                x, y = normt[rows][:, pred[k]], normt[rows][:, targ[k]]
                loss = loss_fn(name.split("+")[0], y, forward(x, masks, in_mask))
                grads = torch.autograd.grad(loss, params)
                losses.append(float(loss))
                step = t + 1
                alpha = lr_ * np.sqrt(1.0 - b2_ ** step) / (1.0 - b1_ ** step)
                with torch.no_grad():
                    for p, gq, mm, vv in zip(params, grads, m, v):
                        mm += (gq - mm) * (1.0 - b1_)
                        vv += (gq * gq - vv) * (1.0 - b2_)
                        p -= alpha * mm / (torch.sqrt(vv) + eps_)
            out["%s/loss_%d" % (name, k)] = np.array(losses)
            with torch.no_grad():
                out["%s/predict_%d" % (name, k)] = forward(normt[:, pred[k]], None).numpy()
            for l in range(3):
                out["%s/W_%d_%d" % (name, k, l)] = params[2 * l].detach().numpy()
                out["%s/b_%d_%d" % (name, k, l)] = params[2 * l + 1].detach().numpy()
                out["%s/vW_%d_%d" % (name, k, l)] = v[2 * l].numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
