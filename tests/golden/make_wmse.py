"""Pin the loss of the hot path to the REFERENCE's own wMSE (deepimpute/multinet.py:36-41) -- the one piece of the NN
arithmetic that can run here without TensorFlow.

Run HERE only (the build container, where /root/reference exists):
    python tests/golden/make_wmse.py
It imports the reference's deepimpute/multinet.py with `tensorflow` replaced by a numpy-backed stub that implements exactly
the four names wMSE touches (tf.cast, tf.float32, tf.square, tf.reduce_mean; float32 arithmetic, as TF would run it on
float32 tensors), calls the reference's wMSE -- binary=False and binary=True -- on seeded (y_true, y_pred) batches and stores
INPUTS and OBSERVED OUTPUTS in tests/golden/kat_wmse.npz:
  * a small one-sub-net problem (matrix, predictor / target columns, batch rows incl. a partial batch, Glorot seed);
    y_true = the targets of the batch rows, y_pred = the sub-net's predictions at its initial weights as the CPU oracle
    computes them (the tests check that the engine under test reproduces y_pred before they compare losses);
  * two free-standing (y_true, y_pred) pairs with their losses;
  * (round 4) dwMSE/dy_pred of the two batches, [b][O] float64, by CENTRAL DIFFERENCES OF THE REFERENCE'S FUNCTION evaluated in
    float64 (the same stub, which then keeps float64: the loss is quadratic in y_pred, so the difference quotient is exact up
    to ~1e-12) -- the gradient the optimiser step starts from, pinned to reference-held code rather than to a restatement.
Nothing of the reference's source is stored; the reference never travels to the GPU box.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def install_tf_stub():
    tf = types.ModuleType("tensorflow")
    tf.float32 = np.float32
    tf.cast = lambda x, dtype: np.asarray(x).astype(dtype)
    tf.square = lambda x: np.square(np.asarray(x))
    # exact mean, rounded once into the dtype of the operand (float32 tensors -> float32, as TF; float64 stays float64)
    tf.reduce_mean = lambda x: np.mean(np.asarray(x), dtype=np.float64).astype(np.asarray(x).dtype if np.asarray(x).dtype == np.float32 else np.float64)
    tf.random = types.SimpleNamespace(set_seed=lambda s: None)
    tf.config = types.SimpleNamespace(threading=types.SimpleNamespace(set_inter_op_parallelism_threads=lambda n: None,
                                                                      set_intra_op_parallelism_threads=lambda n: None))
    mods = {"tensorflow": tf}
    for name in ("tensorflow.keras", "keras", "keras.backend", "keras.models", "keras.layers", "keras.callbacks", "keras.losses"):
        mods[name] = types.ModuleType(name)
    mods["keras"].optimizers = types.SimpleNamespace(Adam=None)
    mods["keras.models"].Model = mods["keras.models"].model_from_json = None
    mods["keras.layers"].Dense = mods["keras.layers"].Dropout = mods["keras.layers"].Input = None
    mods["keras.callbacks"].EarlyStopping = None
    tf.keras = mods["tensorflow.keras"]
    sys.modules.update(mods)


def main():
    if not os.path.isdir(REFERENCE):
        sys.exit("the reference tree is needed to (re)generate this fixture")
    install_tf_stub()
    sys.path.insert(0, REFERENCE)
    sys.path.insert(0, ROOT)
    from deepimpute.multinet import wMSE          # the REFERENCE's loss
    from oracle.dimo import OracleEngine

    rng = np.random.default_rng(20260929)
    n, g, D, H, O = 150, 260, 96, 48, 40
    mu = rng.lognormal(0.5, 1.2, size=g)
    norm = np.log1p(rng.poisson(rng.gamma(2.0, mu / 2.0, size=(n, g)))).astype(np.float32)
    pred = rng.choice(g, D, replace=False).astype(np.int32)
    targ = rng.choice(np.setdiff1d(np.arange(g), pred), O, replace=False).astype(np.int32)
    out = dict(norm=norm, pred=pred, targ=targ, H=H, O=O, seed=np.int64(4711))
    eng = OracleEngine([D], H, O, batch_size=64, dropout_rate=0.0, learning_rate=0.0, seed=4711)
    eng.set_matrix(norm)
    eng.set_indices(0, pred, targ)
    eng.gather(True)
    eng.set_split(np.arange(n - 10, dtype=np.int32), np.arange(n - 10, n, dtype=np.int32))
    eng.init_weights()
    for i, b in enumerate((64, 23)):              # a full and a partial batch
        rows = rng.choice(n, b, replace=False).astype(np.int32)
        y_true = norm[rows][:, targ]
        y_pred = eng.predict(rows)
        assert y_true.dtype == y_pred.dtype == np.float32
        out["rows_%d" % i], out["y_pred_%d" % i] = rows, y_pred
        out["wmse_%d" % i] = np.float32(wMSE(y_true, y_pred))
        out["wmse_binary_%d" % i] = np.float32(wMSE(y_true, y_pred, binary=True))
        yt64, yp64, step = y_true.astype(np.float64), y_pred.astype(np.float64), 1e-4
        for tag, binary in (("dwmse_%d", False), ("dwmse_binary_%d", True)):
            grad = np.empty(yp64.shape, np.float64)
            for idx in np.ndindex(*yp64.shape):
                up, dn = yp64.copy(), yp64.copy()
                up[idx] += step
                dn[idx] -= step
                grad[idx] = (float(wMSE(yt64, up, binary=binary)) - float(wMSE(yt64, dn, binary=binary))) / (2 * step)
            out[tag % i] = grad
    eng.close()
    for i in range(2):                            # free-standing pairs (any restatement of the loss can be checked on these)
        yt = np.log1p(rng.poisson(1.3, size=(37, 29))).astype(np.float32)
        yp = np.log1p(np.exp(rng.normal(0.2, 1.0, size=(37, 29)))).astype(np.float32)
        out["free_true_%d" % i], out["free_pred_%d" % i] = yt, yp
        out["free_wmse_%d" % i] = np.float32(wMSE(yt, yp))
        out["free_wmse_binary_%d" % i] = np.float32(wMSE(yt, yp, binary=True))
    np.savez_compressed(os.path.join(HERE, "kat_wmse.npz"), **out)
    print("wrote kat_wmse.npz:", {k: float(v) for k, v in out.items() if k.startswith(("wmse", "free_wmse"))})


if __name__ == "__main__":
    main()
