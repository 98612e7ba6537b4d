"""Generate tests/golden/kat_epochs.npz: a three-epoch training trajectory of a tiny MultiNet.

Run HERE (CPU): `python tests/golden/make_epochs.py`.

Second, independent pin of the CPU oracle, one level above tests/golden/make_kat.py: that file pins
single optimiser steps with injected batches and masks; this one pins the EPOCH loop of the
reference's `model.fit` (deepimpute/multinet.py:238-246) -- one permutation per epoch shared by all
sub-nets, batches of B with a partial last batch, Keras' running mean of the batch losses weighted by
batch size, validation after every epoch, Adam's step counter running across epochs -- and the
injectable random streams of include/dimn_rng.h, which are restated here from their definition in
plain numpy (Philox4x32-10 as published by Salmon et al., SC'11; checked below against the
Random123 known-answer vectors), NOT by calling the C header.  Gradients come from torch.autograd
in float64; Adam is the Keras/TF form (eps outside the bias correction).  The npz holds inputs and
expected outputs only.
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_epochs.npz")
M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """ctr: 4 uint32, key: 2 uint32 -> 4 uint32 (Random123 philox4x32-10)."""
    c0, c1, c2, c3 = (int(x) for x in ctr)
    k0, k1 = (int(x) for x in key)
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def u01(x):
    return np.float32(x >> 8) * np.float32(1.0 / 16777216.0)


def dropout_keep(seed, kg, epoch, step, b_act, H, rate):
    """keep[b][h] of sub-net kg: element e = b*H + h, 4 elements per Philox block, keep <=> u >= rate."""
    keep = np.zeros((b_act, H), np.uint8)
    for e in range(b_act * H):
        r = philox4x32_10((e >> 2, step, epoch, (2 << 24) | kg), (seed & M32, seed >> 32))
        keep[e // H, e % H] = u01(r[e & 3]) >= np.float32(rate)
    return keep


def epoch_permutation(seed, epoch, n):
    """Fisher-Yates from the back, word i of the permutation stream drives the i-th swap."""
    perm = list(range(n))
    for i in range(n - 1, 0, -1):
        w = n - 1 - i
        r = philox4x32_10((w >> 2, epoch, 0, 3 << 24), (seed & M32, seed >> 32))[w & 3]
        j = (r * (i + 1)) >> 32
        perm[i], perm[j] = perm[j], perm[i]
    return np.array(perm, np.int32)


def glorot(seed, kg, layer, fan_in, fan_out):
    limit = np.float32(np.sqrt(6.0 / (fan_in + fan_out)))
    w = np.empty(fan_in * fan_out, np.float32)
    for e in range(w.size):
        r = philox4x32_10((e >> 2, layer, 0, (1 << 24) | kg), (seed & M32, seed >> 32))
        w[e] = (np.float32(2.0) * u01(r[e & 3]) - np.float32(1.0)) * limit
    return w.reshape(fan_in, fan_out)


def main():
    # Random123 known-answer vectors (kat_vectors, philox4x32 10 rounds)
    assert philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert philox4x32_10((M32,) * 4, (M32, M32)) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)

    rng = np.random.default_rng(77)
    n, g = 150, 90
    H, O, B = 24, 16, 32
    Ds = [20, 17]
    K = len(Ds)
    p, lr, b1c, b2c, eps = 0.25, 2e-3, 0.9, 0.999, 1e-7
    seed, subnet_offset, epochs = 4242, 3, 3            # global sub-net indices 3, 4 key the streams

    lam = rng.lognormal(0.5, 1.2, size=g)
    norm = np.log1p(rng.poisson(rng.gamma(2.0, lam / 2.0, size=(n, g)))).astype(np.float32)
    pred = [rng.choice(g, D, replace=False).astype(np.int32) for D in Ds]
    targ = [rng.choice(g, O, replace=False).astype(np.int32) for _ in Ds]
    val_rows = np.sort(rng.choice(n, 23, replace=False)).astype(np.int32)
    train_rows = np.setdiff1d(np.arange(n, dtype=np.int32), val_rows).astype(np.int32)   # 127 rows: 3 x 32 + 31
    n_tr = train_rows.size

    T = torch.float64
    normt = torch.tensor(norm, dtype=T)
    lr_, b1_, b2_, eps_ = (float(np.float32(x)) for x in (lr, b1c, b2c, eps))
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    out = dict(norm=norm, H=H, O=O, B=B, p=np.float32(p), lr=np.float32(lr), beta1=np.float32(b1c), beta2=np.float32(b2c),
               eps=np.float32(eps), Ds=np.array(Ds, np.int32), seed=np.int64(seed), subnet_offset=np.int32(subnet_offset),
               epochs=np.int32(epochs), val_rows=val_rows, train_rows=train_rows)

    perms = [epoch_permutation(seed, e, n_tr) for e in range(epochs)]
    out["perms"] = np.stack(perms)
    for k in range(K):
        kg = subnet_offset + k
        init = [glorot(seed, kg, 0, Ds[k], H), np.zeros(H, np.float32), glorot(seed, kg, 1, H, O), np.zeros(O, np.float32)]
        for nm, a in zip(("W1", "b1", "W2", "b2"), init):
            out["init_%s_%d" % (nm, k)] = a
        params = [torch.tensor(a, dtype=T, requires_grad=True) for a in init]
        m = [torch.zeros_like(q) for q in params]
        v = [torch.zeros_like(q) for q in params]
        X = normt[:, torch.tensor(pred[k].astype(np.int64))]
        Y = normt[:, torch.tensor(targ[k].astype(np.int64))]

        def forward(rows, keep=None):
            a = torch.relu(X[rows] @ params[0] + params[1])
            if keep is not None:
                a = a * torch.tensor(keep, dtype=T) * scale
            return torch.nn.functional.softplus(a @ params[2] + params[3])

        t, tr_hist, va_hist = 0, [], []
        for e in range(epochs):
            acc = 0.0
            for step, i0 in enumerate(range(0, n_tr, B)):
                rows = train_rows[perms[e][i0:i0 + B]].astype(np.int64)
                keep = dropout_keep(seed, kg, e, step, len(rows), H, p)
                yh, y = forward(rows, keep), Y[rows]
                loss = (y * (y - yh) ** 2).mean()                       # wMSE, multinet.py:36-41
                for q in params:
                    q.grad = None
                loss.backward()
                t += 1
                alpha = lr_ * np.sqrt(1.0 - b2_ ** t) / (1.0 - b1_ ** t)
                with torch.no_grad():
                    for q, mq, vq in zip(params, m, v):
                        mq += (q.grad - mq) * (1.0 - b1_)
                        vq += (q.grad * q.grad - vq) * (1.0 - b2_)
                        q -= alpha * mq / (vq.sqrt() + eps_)
                acc += float(loss.detach()) * len(rows)                          # Keras' size-weighted running mean
            tr_hist.append(acc / n_tr)
            with torch.no_grad():
                yv = Y[val_rows.astype(np.int64)]
                va_hist.append(float((yv * (yv - forward(val_rows.astype(np.int64))) ** 2).mean()))
        out["train_loss_%d" % k] = np.array(tr_hist)
        out["val_loss_%d" % k] = np.array(va_hist)
        with torch.no_grad():
            out["predict_%d" % k] = forward(np.arange(n)).numpy()
        for nm, q, mq, vq in zip(("W1", "b1", "W2", "b2"), params, m, v):
            out["out_%s_%d" % (nm, k)] = q.detach().numpy()
            out["m_%s_%d" % (nm, k)] = mq.numpy()
            out["v_%s_%d" % (nm, k)] = vq.numpy()
        out["pred%d" % k], out["targ%d" % k] = pred[k], targ[k]
        out["steps"] = np.int64(t)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
