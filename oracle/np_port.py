"""BLAS-backed CPU port of the MultiNet hot path.  TEST / BASELINE INFRASTRUCTURE ONLY
(imported by tests/ and bench.py's cpu_baseline leg, never by deepimpute_amd).

Same algorithm as oracle/dimo.c (the S1-S13 restatement of the reference's Keras calls,
deepimpute/multinet.py:36-41, 126-167, 238-244, 278), but every contraction is a numpy matmul
(OpenBLAS sgemm) and the K sub-networks run concurrently on a thread pool -- the shape of what
Keras/TensorFlow does on a CPU (Eigen contractions + inter-op parallelism over the K branches,
multinet.py:222-223).  It exists so that the timed CPU baseline is a competent one; dimo.c
stays the plain-loop checker and tests/test_np_port.py pins this file against it.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _softplus(z):
    thr = np.float32(13.942385)
    mid = np.log1p(np.exp(np.clip(z, -thr, thr)))
    return np.where(z > thr, z, np.where(z < -thr, np.exp(np.minimum(z, 0)), mid)).astype(np.float32)


class NumpyPort:
    def __init__(self, D, hidden, out_dim, batch_size=64, dropout_rate=0.2, learning_rate=1e-4, beta1=0.9,
                 beta2=0.999, eps=1e-7, seed=1234, threads=None, **_):
        self.D, self.K, self.H, self.O, self.B = list(D), len(D), hidden, out_dim, batch_size
        self.p, self.lr = np.float32(dropout_rate), np.float32(learning_rate)
        self.b1c, self.b2c, self.eps = np.float32(beta1), np.float32(beta2), np.float32(eps)
        self.t = 0
        self.rng = np.random.default_rng(seed)
        self.pool = ThreadPoolExecutor(threads or min(self.K, os.cpu_count() or 1))
        self.W = [None] * self.K          # [W1, b1, W2, b2] per sub-net, Keras layout
        self.M = [None] * self.K
        self.V = [None] * self.K

    # same data interface as the engines
    def set_matrix(self, norm):
        self.norm = np.ascontiguousarray(norm, np.float32)

    def set_indices(self, k, pred, targ):
        if not hasattr(self, "pred"):
            self.pred, self.targ = [None] * self.K, [None] * self.K
        self.pred[k], self.targ[k] = np.asarray(pred, np.int64), np.asarray(targ, np.int64)

    def gather(self, with_targets=True):
        # the reference materialises X_k / Y_k once (multinet.py:231-235)
        self.X = [np.ascontiguousarray(self.norm[:, p]) for p in self.pred]
        if with_targets:
            self.Y = [np.ascontiguousarray(self.norm[:, t]) for t in self.targ]

    def set_weights(self, k, W1, b1, W2, b2):
        self.W[k] = [np.array(a, np.float32) for a in (W1, b1, W2, b2)]
        self.M[k] = [np.zeros_like(a) for a in self.W[k]]
        self.V[k] = [np.zeros_like(a) for a in self.W[k]]

    def init_weights(self):
        for k in range(self.K):
            l1 = np.sqrt(6.0 / (self.D[k] + self.H)); l2 = np.sqrt(6.0 / (self.H + self.O))
            self.set_weights(k, self.rng.uniform(-l1, l1, (self.D[k], self.H)), np.zeros(self.H),
                             self.rng.uniform(-l2, l2, (self.H, self.O)), np.zeros(self.O))
        self.t = 0

    def get_weights(self, k):
        return tuple(self.W[k])

    def _step_one(self, k, rows, keep, alpha):
        W1, b1, W2, b2 = self.W[k]
        x, y = self.X[k][rows], self.Y[k][rows]
        scale = np.float32(1.0) / (np.float32(1.0) - self.p)
        a = x @ W1 + b1
        gate = (a > 0) & keep
        dd = np.where(gate, a * scale, np.float32(0)).astype(np.float32)
        z = dd @ W2 + b2
        yh = _softplus(z)
        e = y - yh
        inv_n = np.float32(1.0 / (rows.size * self.O))
        loss = float(np.sum(y * e * e, dtype=np.float64)) / (rows.size * self.O)
        dz = (np.float32(-2) * y * e * inv_n / (np.float32(1) + np.exp(-z))).astype(np.float32)
        grads = [None, None, dd.T @ dz, dz.sum(0)]
        dA = np.where(gate, (dz @ W2.T) * scale, np.float32(0)).astype(np.float32)
        grads[0], grads[1] = x.T @ dA, dA.sum(0)
        omb1, omb2 = np.float32(1) - self.b1c, np.float32(1) - self.b2c
        for w, m, v, g in zip(self.W[k], self.M[k], self.V[k], grads):     # Keras-form Adam, in place
            m += (g - m) * omb1
            v += (g * g - v) * omb2
            w -= (m * alpha) / (np.sqrt(v) + self.eps)
        return loss

    def train_step(self, rows, keep_mask=None, **_):
        rows = np.asarray(rows, np.int64)
        self.t += 1
        alpha = np.float32(float(self.lr) * np.sqrt(1.0 - float(self.b2c) ** self.t) / (1.0 - float(self.b1c) ** self.t))
        if keep_mask is None:
            keep_mask = self.rng.random((self.K, rows.size, self.H), dtype=np.float32) >= self.p
        keep_mask = np.asarray(keep_mask).astype(bool)
        return np.array(list(self.pool.map(lambda k: self._step_one(k, rows, keep_mask[k], alpha), range(self.K))), np.float32)

    def _forward(self, k, rows):
        W1, b1, W2, b2 = self.W[k]
        x = self.X[k] if rows is None else self.X[k][rows]
        return _softplus(np.maximum(x @ W1 + b1, 0) @ W2 + b2)

    def predict(self, rows=None):
        rows = None if rows is None else np.asarray(rows, np.int64)
        return np.hstack(list(self.pool.map(lambda k: self._forward(k, rows), range(self.K))))

    def close(self):
        self.pool.shutdown()
