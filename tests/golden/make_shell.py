"""Capture fixtures that pin deepimpute_amd.multinet.MultiNet's host shell against the REFERENCE.

Run HERE only (the build container, where /root/reference exists):
    python tests/golden/make_shell.py
It imports the reference's deepimpute/multinet.py with TensorFlow/Keras replaced by in-memory
recording stubs (TF is not installable in this image), runs MultiNet.fit()/predict() on small
seeded synthetic count matrices and stores INPUTS and OBSERVED OUTPUTS in
tests/golden/shell_cases.npz + shell_cases.json:
  - genes/targets/predictors chosen by the reference, the validation cells, the arguments it
    hands to Dense/Dropout/Adam/EarlyStopping/model.fit, shapes of X/Y,
  - predict() post-processing output for a known, deterministic fake prediction.
Nothing of the reference's source is stored; the reference never travels to the GPU box.
"""
import json
import os
import sys
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
LOG = {}


def fake_prediction(x, k, out_dim):
    """Deterministic stand-in for a sub-network's output, shared with tests/test_shell.py."""
    base = x.mean(axis=1, keepdims=True).astype(np.float32) * np.float32(0.5)
    ramp = (np.arange(out_dim, dtype=np.float32) % 7) * np.float32(0.03) + np.float32(0.01 * k)
    return (base + ramp[None, :]).astype(np.float32)


def install_stubs():
    class Tensor:
        def __init__(self, dim):
            self.dim = dim

    class Layer:
        def __init__(self, kind, *a, **kw):
            LOG.setdefault("layers", []).append([kind, [repr(x) for x in a], {k: repr(v) for k, v in kw.items()}])
            self.kind, self.a, self.kw = kind, a, kw

        def __call__(self, t):
            return Tensor(self.a[0] if self.kind == "Dense" else t.dim)

    class History:
        def __init__(self):
            self.history = {"loss": [3.0, 2.0, 1.0]}

    class Model:
        last = None

        def __init__(self, inputs, outputs):
            self.inputs, self.outputs = inputs, outputs
            Model.last = self

        def compile(self, optimizer=None, loss=None):
            LOG["compile"] = {"optimizer": optimizer, "loss": getattr(loss, "__name__", str(loss))}

        def fit(self, X, Y, validation_data=None, epochs=None, batch_size=None, callbacks=None, verbose=None):
            LOG["fit"] = {"epochs": epochs, "batch_size": batch_size, "verbose": verbose,
                          "callbacks": [c for c in callbacks],
                          "x_shapes": [list(x.shape) for x in X], "y_shapes": [list(y.shape) for y in Y],
                          "xv_shapes": [list(x.shape) for x in validation_data[0]],
                          "dtypes": sorted({str(x.dtype) for x in list(X) + list(Y)})}
            LOG["_arrays"] = {"X": X, "Y": Y, "Xv": validation_data[0], "Yv": validation_data[1]}
            return History()

        def predict(self, X):
            outs = [fake_prediction(np.asarray(x), k, self.outputs[k].dim) for k, x in enumerate(X)]
            return outs if len(outs) > 1 else outs[0]

        def to_json(self):
            return "{}"

        def save_weights(self, path):
            open(path, "w").close()

        def load_weights(self, path):
            pass

    tf = types.ModuleType("tensorflow")
    tf.random = types.SimpleNamespace(set_seed=lambda s: LOG.__setitem__("tf_seed", s))
    tf.config = types.SimpleNamespace(threading=types.SimpleNamespace(
        set_inter_op_parallelism_threads=lambda n: LOG.__setitem__("inter", n),
        set_intra_op_parallelism_threads=lambda n: LOG.__setitem__("intra", n)))
    tf.cast = lambda x, t: x
    tf.float32 = np.float32
    tfk = types.ModuleType("tensorflow.keras")
    keras = types.ModuleType("keras")
    keras.backend = types.ModuleType("keras.backend")
    keras.models = types.ModuleType("keras.models")
    keras.layers = types.ModuleType("keras.layers")
    keras.callbacks = types.ModuleType("keras.callbacks")
    keras.losses = types.ModuleType("keras.losses")
    keras.optimizers = types.SimpleNamespace(Adam=lambda **kw: {"Adam": kw})
    keras.models.Model = Model
    keras.models.model_from_json = lambda s: Model.last
    keras.layers.Dense = lambda *a, **kw: Layer("Dense", *a, **kw)
    keras.layers.Dropout = lambda *a, **kw: Layer("Dropout", *a, **kw)
    keras.layers.Input = lambda shape=None: Tensor(shape[0])
    keras.callbacks.EarlyStopping = lambda **kw: {"EarlyStopping": kw}
    tf.keras = tfk
    for name, mod in [("tensorflow", tf), ("tensorflow.keras", tfk), ("keras", keras), ("keras.backend", keras.backend),
                      ("keras.models", keras.models), ("keras.layers", keras.layers),
                      ("keras.callbacks", keras.callbacks), ("keras.losses", keras.losses)]:
        sys.modules[name] = mod


def synth_raw(n, g, seed):
    rng = np.random.default_rng(seed)
    mu = rng.lognormal(0.5, 1.2, size=g)
    counts = rng.poisson(rng.gamma(2.0, mu / 2.0, size=(n, g))).astype(np.float64)
    return pd.DataFrame(counts, index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])


CASES = [
    # name, data (n, g, seed), constructor kwargs, fit kwargs
    ("default64", (160, 420, 1), dict(sub_outputdim=64, seed=123, ncores=2, verbose=0), dict()),
    ("limit_ntop3", (120, 300, 2), dict(sub_outputdim=32, seed=7, ncores=1, verbose=0,
                                        architecture=[{"type": "dense", "activation": "relu", "neurons": 24},
                                                      {"type": "dropout", "activation": "dropout", "rate": 0.3}]),
     dict(NN_lim=100, ntop=3)),
    ("progressive", (100, 260, 3), dict(sub_outputdim=32, seed=99, ncores=1, verbose=0), dict(NN_lim=64, mode="progressive")),
    ("gene_list", (110, 280, 4), dict(sub_outputdim=64, seed=5, ncores=1, verbose=0),
     dict(genes_to_impute=["g%d" % j for j in range(3, 53)])),
    ("subset", (140, 240, 5), dict(sub_outputdim=32, seed=11, ncores=1, verbose=0), dict(cell_subset=0.8, NN_lim=40)),
]


def main():
    if not os.path.isdir(REFERENCE):
        sys.exit("the reference tree is needed to (re)generate these fixtures")
    install_stubs()
    sys.path.insert(0, REFERENCE)
    import warnings
    warnings.simplefilter("ignore")
    from deepimpute.multinet import MultiNet  # the REFERENCE implementation

    arrays, meta = {}, {}
    for name, (n, g, dseed), ctor, fitkw in CASES:
        LOG.clear()
        raw = synth_raw(n, g, dseed)
        outdir = os.path.join("/tmp", "dimn_shell_" + name)
        net = MultiNet(output_prefix=outdir, **ctor)
        net.fit(raw, **fitkw)
        fitlog = dict(LOG["fit"])
        arrs = LOG["_arrays"]
        K = len(net.predictors)
        # the split depends only on (seed, index labels) right after the second np.random.seed
        # (multinet.py:219,228); the rows actually fitted are recovered from the frames the
        # reference built, which also covers the cell_subset path
        norm_used_index = None
        np.random.seed(ctor["seed"])
        # reproduce `raw` as the reference saw it (cell_subset re-samples rows first)
        raw_used = raw
        if fitkw.get("cell_subset", 1) != 1:
            np.random.seed(ctor["seed"])
            raw_used = raw.sample(frac=fitkw["cell_subset"])
        np.random.seed(ctor["seed"])
        test_cells = np.random.choice(raw_used.index, int(0.05 * raw_used.shape[0]), replace=False)
        train_cells = np.setdiff1d(raw_used.index, test_cells)
        norm = np.log1p(raw_used).astype(np.float32)
        assert np.array_equal(arrs["Xv"][0], norm.loc[test_cells, net.predictors[0]].values)
        assert np.array_equal(arrs["X"][K - 1], norm.loc[train_cells, net.predictors[K - 1]].values)
        assert np.array_equal(arrs["Y"][0], norm.loc[train_cells, net.targets[0]].values)

        col = {c: i for i, c in enumerate(raw.columns)}
        row = {r: i for i, r in enumerate(raw.index)}
        arrays[name + "/raw"] = raw.values.astype(np.int32)
        arrays[name + "/targets"] = np.array([[col[x] for x in t] for t in net.targets], np.int32)
        for k in range(K):
            arrays["%s/pred%d" % (name, k)] = np.array([col[x] for x in net.predictors[k]], np.int32)
        arrays[name + "/test_cells"] = np.array([row[x] for x in test_cells], np.int32)
        arrays[name + "/train_cells"] = np.array([row[x] for x in train_cells], np.int32)
        arrays[name + "/used_rows"] = np.array([row[x] for x in raw_used.index], np.int32)
        for policy in ("restore", "max"):
            arrays["%s/imputed_%s" % (name, policy)] = net.predict(raw, policy=policy).values
        only = net.predict(raw, imputed_only=True, policy="restore")
        arrays[name + "/imputed_only_cols"] = np.array([col[x] for x in only.columns], np.int32)
        arrays[name + "/imputed_only"] = only.values
        arrays[name + "/test_corr_mse"] = np.array([net.test_metrics["correlation"], net.test_metrics["MSE"]], np.float64)
        meta[name] = {"n": n, "g": g, "ctor": ctor, "fit": fitkw, "K": K,
                      "layers": LOG["layers"], "compile": LOG["compile"], "fit_call": fitlog,
                      "tf_seed": LOG.get("tf_seed"), "threads": [LOG.get("inter"), LOG.get("intra")],
                      "trained_epochs": net.trained_epochs}
    np.savez_compressed(os.path.join(HERE, "shell_cases.npz"), **arrays)
    with open(os.path.join(HERE, "shell_cases.json"), "w") as f:
        json.dump(meta, f, indent=1, default=str)
    print("wrote shell_cases.npz (%d bytes), %d cases" % (os.path.getsize(os.path.join(HERE, "shell_cases.npz")), len(CASES)))


if __name__ == "__main__":
    main()
