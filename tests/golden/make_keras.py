"""Generate tests/golden/kat_keras.npz: the SAME problem as kat_steps.npz (make_kat.py) run through real Keras.

Run wherever `tensorflow` (>= 2.0, the reference's requirement: setup.py:25-27) imports:
    python tests/golden/make_keras.py
It builds the reference's model -- one Keras functional Model with K inputs and K outputs, each branch
Dense(H, relu) -> Dropout -> Dense(O, softplus), compiled with keras.optimizers.Adam(lr) and the reference's wMSE
(deepimpute/multinet.py:36-41, 126-167) -- loads the weights of kat_steps.npz, and runs the three batches of
kat_steps.npz through model.train_on_batch with the dropout rate set to 0 (Keras's dropout stream cannot be
injected; rate 0 makes the layer an identity, so the step is deterministic), then model.predict on all rows.
Outputs: per-step per-branch losses, final weights, predictions.  tests/test_oracle_kat.py::test_oracle_matches_keras
compares oracle/dimo.c (keep-mask of ones) against this file when it exists.

STATUS (round 2): TensorFlow/Keras cannot be installed in the build container (no network: `pip install
tensorflow-cpu` -> "No matching distribution found") nor on the GPU boxes, so this script has NOT been run and
kat_keras.npz is NOT committed: the NN arithmetic of the oracle stays pinned to torch-fp64 autograd only
("parity unpinned" against Keras).
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main_oracle(out_path):
    """The SAME file layout from the CPU oracle: a stand-in that lets the consumer (tests/helpers.check_against_keras_file) run end
    to end without TensorFlow.  It pins nothing -- the oracle is compared with itself -- and says so in `source`."""
    import sys
    root = os.path.dirname(os.path.dirname(HERE))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import kat_engine, load_kat
    from oracle.dimo import OracleEngine
    kat = load_kat()
    K, O = len(kat["Ds"]), int(kat["O"])
    eng = kat_engine(OracleEngine, kat, dropout_rate=0.0)
    out = {"source": np.array("oracle/dimo.c stand-in (plumbing check, NOT Keras)")}
    out["loss"] = np.array([eng.train_step(kat["rows_%d" % t]) for t in range(3)], np.float64)
    pred = eng.predict()
    for k in range(K):
        W1, b1, W2, b2 = eng.get_weights(k)
        out.update({"W1_%d" % k: W1, "b1_%d" % k: b1, "W2_%d" % k: W2, "b2_%d" % k: b2, "predict_%d" % k: pred[:, k * O:(k + 1) * O]})
    eng.close()
    np.savez_compressed(out_path, **out)


def main(backend="keras", out_path=None):
    if backend == "oracle":
        return main_oracle(out_path)
    import tensorflow as tf
    from tensorflow import keras
    from tensorflow.keras import backend as Kb
    from tensorflow.keras.layers import Dense, Dropout, Input

    kat = np.load(os.path.join(HERE, "kat_steps.npz"))
    Ds = [int(d) for d in kat["Ds"]]
    K, H, O = len(Ds), int(kat["H"]), int(kat["O"])
    norm = kat["norm"]

    def wMSE(y_true, y_pred):                       # deepimpute/multinet.py:36-41 (binary=False)
        return tf.reduce_mean(y_true * tf.square(y_true - y_pred))

    inputs = [Input(shape=(D,)) for D in Ds]
    outputs = []
    for x in inputs:                                # multinet.py:132-146
        x = Dense(H, activation="relu")(x)
        x = Dropout(0.0)(x)
        outputs.append(Dense(O, activation="softplus")(x))
    model = keras.Model(inputs=inputs, outputs=outputs)
    model.compile(optimizer=keras.optimizers.Adam(learning_rate=float(kat["lr"])), loss=wMSE)      # multinet.py:164-165
    dense = [l for l in model.layers if isinstance(l, Dense)]
    # functional-API layer order: all first Dense layers, then all output layers (K branches created in a loop)
    firsts = [l for l in dense if l.units == H][:K]
    lasts = [l for l in dense if l.units == O][-K:]
    for k in range(K):
        firsts[k].set_weights([kat["W1_%d" % k], kat["b1_%d" % k]])
        lasts[k].set_weights([kat["W2_%d" % k], kat["b2_%d" % k]])

    out = {"tf_version": np.array(tf.__version__), "keras_version": np.array(keras.__version__)}
    losses = np.zeros((3, K))
    for t in range(3):
        rows = kat["rows_%d" % t]
        X = [norm[np.ix_(rows, kat["pred%d" % k])] for k in range(K)]
        Y = [norm[np.ix_(rows, kat["targ%d" % k])] for k in range(K)]
        res = model.train_on_batch(X, Y)            # [total, branch 0, ..., branch K-1]
        losses[t] = np.asarray(res)[1:1 + K]
    out["loss"] = losses
    Xall = [norm[:, kat["pred%d" % k]] for k in range(K)]
    pred = model.predict(Xall, verbose=0)
    for k in range(K):
        W1, b1 = firsts[k].get_weights()
        W2, b2 = lasts[k].get_weights()
        out.update({"W1_%d" % k: W1, "b1_%d" % k: b1, "W2_%d" % k: W2, "b2_%d" % k: b2, "predict_%d" % k: pred[k]})
    np.savez_compressed(out_path or os.path.join(HERE, "kat_keras.npz"), **out)
    print("wrote kat_keras.npz (tensorflow %s)" % tf.__version__)


if __name__ == "__main__":
    main()
