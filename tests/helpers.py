"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def multinet_with(engine_factory, **kw):
    """A MultiNet whose build() seam constructs `engine_factory` engines (and `engine_factory.general` ones for what the tuned
    kernels do not take) -- test infrastructure: the product class binds libdimn.so and has no such hook.  Planning on the
    device from the resident counts is a feature of the HIP engine, so it is off here."""
    from deepimpute_amd.multinet import MultiNet

    class Injected(MultiNet):
        _device_planning = False

        def _engine_classes(self):
            return engine_factory, getattr(engine_factory, "general", None)
    return Injected(**kw)


def load_kat():
    return np.load(os.path.join(GOLDEN, "kat_steps.npz"))


def kat_engine(cls, kat, dropout_rate=None, **kw):
    """Build an engine of class `cls` loaded with the KAT problem (weights injected)."""
    Ds = [int(d) for d in kat["Ds"]]
    eng = cls(Ds, int(kat["H"]), int(kat["O"]), batch_size=int(kat["B"]),
              dropout_rate=float(kat["p"]) if dropout_rate is None else dropout_rate, learning_rate=float(kat["lr"]),
              beta1=float(kat["beta1"]), beta2=float(kat["beta2"]), eps=float(kat["eps"]),
              seed=7, **kw)
    eng.set_matrix(kat["norm"])
    for k in range(len(Ds)):
        eng.set_indices(k, kat["pred%d" % k], kat["targ%d" % k])
    eng.gather(True)
    n = kat["norm"].shape[0]
    val = kat["val_rows"]
    train = np.setdiff1d(np.arange(n, dtype=np.int32), val).astype(np.int32)
    eng.set_split(train, val)
    eng.reset_optimizer()
    for k in range(len(Ds)):
        eng.set_weights(k, kat["W1_%d" % k], kat["b1_%d" % k], kat["W2_%d" % k], kat["b2_%d" % k])
    return eng


def run_kat_steps(eng, kat):
    losses = []
    for t in range(3):
        losses.append(eng.train_step(kat["rows_%d" % t], keep_mask=kat["mask_%d" % t]))
    return np.array(losses)  # [3][K]


def check_kat(eng, kat, rtol, atol):
    """Run the three KAT steps on `eng` and compare everything with the stored answers."""
    K = len(kat["Ds"])
    losses = run_kat_steps(eng, kat)
    for k in range(K):
        np.testing.assert_allclose(losses[:, k], kat["loss_%d" % k], rtol=rtol, atol=atol)
    assert eng.step_count() == 3
    vl = eng.val_loss()
    pred = eng.predict()
    for k in range(K):
        W = eng.get_weights(k)
        M = eng.get_adam_state(k, 0)
        V = eng.get_adam_state(k, 1)
        for i, name in enumerate(("W1", "b1", "W2", "b2")):
            np.testing.assert_allclose(W[i], kat["out_%s_%d" % (name, k)], rtol=rtol, atol=atol,
                                       err_msg="weights %s k=%d" % (name, k))
            np.testing.assert_allclose(M[i], kat["m_%s_%d" % (name, k)], rtol=rtol, atol=atol * 1e-2,
                                       err_msg="adam m %s k=%d" % (name, k))
            np.testing.assert_allclose(V[i], kat["v_%s_%d" % (name, k)], rtol=rtol, atol=atol * 1e-4,
                                       err_msg="adam v %s k=%d" % (name, k))
        O = int(kat["O"])
        np.testing.assert_allclose(pred[:, k * O:(k + 1) * O], kat["pred_all_%d" % k], rtol=rtol,
                                   atol=atol, err_msg="predict k=%d" % k)
        np.testing.assert_allclose(vl[k], kat["val_loss_%d" % k], rtol=rtol)


def make_problem(n, g, Ds, H, O, seed=0, val_frac=0.1):
    """Seeded synthetic log1p(count) matrix + random predictor/target column lists."""
    rng = np.random.default_rng(seed)
    lam = rng.lognormal(0.5, 1.2, size=g)
    norm = np.log1p(rng.poisson(rng.gamma(2.0, lam / 2.0, size=(n, g)))).astype(np.float32)
    pred = [rng.choice(g, D, replace=False).astype(np.int32) for D in Ds]
    targ = [rng.choice(g, O, replace=O > g).astype(np.int32) for _ in Ds]
    val = np.sort(rng.choice(n, max(1, int(val_frac * n)), replace=False)).astype(np.int32)
    train = np.setdiff1d(np.arange(n, dtype=np.int32), val).astype(np.int32)
    return dict(norm=norm, pred=pred, targ=targ, Ds=list(Ds), H=H, O=O, train=train, val=val)


def load_problem(cls, prob, **kw):
    eng = cls(prob["Ds"], prob["H"], prob["O"], **kw)
    eng.set_matrix(prob["norm"])
    for k in range(len(prob["Ds"])):
        eng.set_indices(k, prob["pred"][k], prob["targ"][k])
    eng.gather(True)
    eng.set_split(prob["train"], prob["val"])
    return eng


def load_epochs():
    return np.load(os.path.join(GOLDEN, "kat_epochs.npz"))


def check_epoch_trajectory(cls, rtol_loss, rtol_w, atol_w, **kw):
    """Three epochs of a tiny MultiNet against tests/golden/kat_epochs.npz (torch fp64 autograd + a numpy
    restatement of the Philox streams; generator: tests/golden/make_epochs.py): initial weights and the
    per-epoch permutations bit-exact, losses / weights / Adam state / predictions within the tolerances."""
    z = load_epochs()
    Ds = [int(d) for d in z["Ds"]]
    K = len(Ds)
    eng = cls(Ds, int(z["H"]), int(z["O"]), batch_size=int(z["B"]), dropout_rate=float(z["p"]),
              learning_rate=float(z["lr"]), beta1=float(z["beta1"]), beta2=float(z["beta2"]), eps=float(z["eps"]),
              seed=int(z["seed"]), subnet_offset=int(z["subnet_offset"]), **kw)
    eng.set_matrix(z["norm"])
    for k in range(K):
        eng.set_indices(k, z["pred%d" % k], z["targ%d" % k])
    eng.gather(True)
    eng.set_split(z["train_rows"], z["val_rows"])
    eng.init_weights()
    for k in range(K):
        for got, name in zip(eng.get_weights(k), ("W1", "b1", "W2", "b2")):
            assert np.array_equal(got, z["init_%s_%d" % (name, k)]), "init %s k=%d" % (name, k)
    for e in range(int(z["epochs"])):
        assert np.array_equal(eng.epoch_permutation(e), z["perms"][e])
        tr = eng.train_epoch(e)                      # library-generated permutation and dropout masks
        va = eng.val_loss()
        for k in range(K):
            np.testing.assert_allclose(tr[k], z["train_loss_%d" % k][e], rtol=rtol_loss, err_msg="train loss epoch %d k=%d" % (e, k))
            np.testing.assert_allclose(va[k], z["val_loss_%d" % k][e], rtol=rtol_loss, err_msg="val loss epoch %d k=%d" % (e, k))
    assert eng.step_count() == int(z["steps"])
    pred = eng.predict()
    O = int(z["O"])
    for k in range(K):
        W, M, V = eng.get_weights(k), eng.get_adam_state(k, 0), eng.get_adam_state(k, 1)
        for i, name in enumerate(("W1", "b1", "W2", "b2")):
            np.testing.assert_allclose(W[i], z["out_%s_%d" % (name, k)], rtol=rtol_w, atol=atol_w, err_msg="%s k=%d" % (name, k))
            np.testing.assert_allclose(M[i], z["m_%s_%d" % (name, k)], rtol=rtol_w, atol=atol_w * 1e-2, err_msg="m %s k=%d" % (name, k))
            np.testing.assert_allclose(V[i], z["v_%s_%d" % (name, k)], rtol=rtol_w, atol=atol_w * 1e-4, err_msg="v %s k=%d" % (name, k))
        np.testing.assert_allclose(pred[:, k * O:(k + 1) * O], z["predict_%d" % k], rtol=rtol_w, atol=atol_w)
    eng.close()


def check_activation_kat(cls, name, rtol, atol, **kw):
    """Two optimiser steps + predict of a tiny sub-net with hidden activation `name` against
    tests/golden/kat_act.npz (torch fp64 autograd; generator tests/golden/make_act.py)."""
    z = np.load(os.path.join(GOLDEN, "kat_act.npz"))
    eng = cls([int(z["D"])], int(z["H"]), int(z["O"]), batch_size=int(z["B"]), dropout_rate=float(z["p"]),
              learning_rate=float(z["lr"]), beta1=float(z["beta1"]), beta2=float(z["beta2"]), eps=float(z["eps"]),
              seed=1, activation=name, **kw)
    eng.set_matrix(z["norm"])
    eng.set_indices(0, z["pred"], z["targ"])
    eng.gather(True)
    n = z["norm"].shape[0]
    eng.set_split(np.arange(n - 8, dtype=np.int32), np.arange(n - 8, n, dtype=np.int32))
    eng.reset_optimizer()
    eng.set_weights(0, z["init_W1"], z["init_b1"], z["init_W2"], z["init_b2"])
    for t in range(2):
        loss = eng.train_step(z["rows_%d" % t], keep_mask=z["mask_%d" % t])
        np.testing.assert_allclose(loss[0], z[name + "/loss"][t], rtol=rtol, err_msg="%s loss step %d" % (name, t))
    for got, nm in zip(eng.get_weights(0), ("W1", "b1", "W2", "b2")):
        np.testing.assert_allclose(got, z["%s/%s" % (name, nm)], rtol=rtol, atol=atol, err_msg="%s %s" % (name, nm))
    np.testing.assert_allclose(eng.predict(), z[name + "/predict"], rtol=rtol, atol=atol, err_msg=name + " predict")
    eng.close()


def check_general_kat(cls, loss, rtol, atol, **kw):
    """Two optimiser steps (batch 100 > 64, then a partial batch) + predict of the two-hidden-layer model of
    tests/golden/kat_general.npz (torch-fp64 autograd, make_general.py) for one loss, on a GeneralEngine class."""
    z = np.load(os.path.join(GOLDEN, "kat_general.npz"))
    Ds = [int(d) for d in z["Ds"]]
    layers = [(int(w), str(a), float(p)) for w, a, p in zip(z["widths"], z["acts"], z["rates"])]
    if loss.endswith("+input_dropout"):                  # a Dropout layer before the first Dense layer: the leading (0, _, rate) entry
        layers = [(0, "linear", float(z["input_dropout_rate"]))] + layers
    eng = cls(Ds, layers, int(z["O"]), batch_size=int(z["B"]), learning_rate=float(z["lr"]), beta1=float(z["beta1"]), beta2=float(z["beta2"]),
              eps=float(z["eps"]), loss=loss.split("+")[0], seed=int(z["seed"]), subnet_offset=int(z["subnet_offset"]), **kw)
    eng.set_matrix(z["norm"])
    for k in range(len(Ds)):
        eng.set_indices(k, z["pred%d" % k], z["targ%d" % k])
    eng.gather(True)
    n = z["norm"].shape[0]
    eng.set_split(np.arange(n - 8, dtype=np.int32), np.arange(n - 8, n, dtype=np.int32))
    for k in range(len(Ds)):
        for l in range(3):
            eng.set_layer_weights(k, l, z["init_W_%d_%d" % (k, l)], z["init_b_%d_%d" % (k, l)])
    for t in range(2):
        got = eng.train_step(z["rows_%d" % t], epoch_key=0, step_key=t)
        for k in range(len(Ds)):
            np.testing.assert_allclose(got[k], z["%s/loss_%d" % (loss, k)][t], rtol=rtol, err_msg="%s loss step %d k=%d" % (loss, t, k))
    assert eng.step_count() == 2
    pred = eng.predict()
    O = int(z["O"])
    for k in range(len(Ds)):
        for l in range(3):
            W, b = eng.get_layer_weights(k, l)
            np.testing.assert_allclose(W, z["%s/W_%d_%d" % (loss, k, l)], rtol=rtol, atol=atol, err_msg="%s W k=%d l=%d" % (loss, k, l))
            np.testing.assert_allclose(b, z["%s/b_%d_%d" % (loss, k, l)], rtol=rtol, atol=atol, err_msg="%s b k=%d l=%d" % (loss, k, l))
            vW, _ = eng.get_layer_weights(k, l, which=2)
            np.testing.assert_allclose(vW, z["%s/vW_%d_%d" % (loss, k, l)], rtol=10 * rtol, atol=atol * 1e-4, err_msg="%s v k=%d l=%d" % (loss, k, l))
        np.testing.assert_allclose(pred[:, k * O:(k + 1) * O], z["%s/predict_%d" % (loss, k)], rtol=rtol, atol=atol, err_msg="%s predict k=%d" % (loss, k))
    eng.close()


def relu_flip_units(hip, orc, k, rtol=1e-3, atol=2e-5):
    """Hidden units of sub-net k whose first-layer column / bias differ between two engines beyond the weight tolerance."""
    Wa, ba = hip.get_weights(k)[:2]
    Wb, bb = orc.get_weights(k)[:2]
    bad_w = ~np.all(np.abs(Wa - Wb) <= atol + rtol * np.abs(Wb), axis=0)
    bad_b = ~(np.abs(ba - bb) <= atol + rtol * np.abs(bb))
    return np.flatnonzero(bad_w | bad_b)


def _round_bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def find_relu_flip_candidates(oracle_cls, norm, pred_k, targ_k, k_global, units, train, steps, H, O, slack=4.0, epochs=1, **kw):
    """The mechanism behind a (rare) disagreement of two fp32 paths that sum a first-layer dot product in different orders:
    relu'(a) is discontinuous at a = 0.  The fp64 oracle replays the first `epochs` epochs (`steps` optimiser steps each) of sub-net
    `k_global` alone (Philox streams are keyed by GLOBAL sub-net index) and reports every (epoch, step, batch position, unit) among
    `units` whose pre-activation lies within the reordering error of an fp32 dot product of zero: |a| <= slack * eps32 *
    (sum_i |x_i w_i| + |b|) -- the places where the SIGN of a is not defined at fp32 precision.
    Records (epoch, step, b, unit, a, bound), closest to zero first."""
    o64 = oracle_cls([len(pred_k)], H, O, fp64=True, subnet_offset=int(k_global), **kw)
    o64.set_matrix(norm)
    o64.set_indices(0, pred_k, targ_k)
    o64.gather(True)
    o64.set_split(train, train[:1])
    o64.init_weights()
    B = o64.B
    units = np.asarray(units)
    bf16 = str(kw.get("precision", "fp32")).lower() in ("bf16", "bfloat16")
    out = []
    for e in range(epochs):
        perm = o64.epoch_permutation(e)
        for t in range(steps):
            rows = train[perm[t * B:(t + 1) * B]]
            W1, b1 = o64.get_weights(0)[:2]
            X = norm[rows][:, pred_k]
            if bf16:                                                   # the arena stores the predictors rounded to nearest even
                X = _round_bf16(X)
            X = X.astype(np.float64)
            w = W1[:, units].astype(np.float64)
            a = X @ w + b1[units].astype(np.float64)
            bound = slack * np.finfo(np.float32).eps * (np.abs(X) @ np.abs(w) + np.abs(b1[units].astype(np.float64)))
            for i, j in zip(*np.nonzero(np.abs(a) <= bound)):
                out.append((e, t, int(i), int(units[j]), float(a[i, j]), float(bound[i, j])))
            o64.train_step(rows, epoch_key=e, step_key=t, want_loss=False)
    o64.close()
    return sorted(out, key=lambda r: abs(r[4]) / r[5])


def oracle_with_inverted_gates(oracle_cls, norm, pred_k, targ_k, k_global, train, val, H, O, inversions, epochs=1, **kw):
    """The fp32 oracle of sub-net `k_global` alone, trained for `epochs` epochs with the listed (epoch, step, b, unit) relu gates taken
    on the other side of zero (oracle/dimo.c dimo_invert_gate: a test instrument for pre-activations at fp32 noise level).
    Returns the engine and the last epoch's training loss."""
    o = oracle_cls([len(pred_k)], H, O, subnet_offset=int(k_global), **kw)
    o.set_matrix(norm)
    o.set_indices(0, pred_k, targ_k)
    o.gather(True)
    o.set_split(train, val)
    o.init_weights()
    for epoch, step, b, unit in inversions:
        o.invert_gate(0, epoch, step, b, unit)
    loss = None
    for e in range(epochs):
        loss = o.train_epoch(e)
    return o, loss


def check_reference_wmse(cls, rtol, **kw):
    """The loss an engine reports for an optimiser step (lr = 0, no dropout) against the REFERENCE's own wMSE
    (deepimpute/multinet.py:36-41, both `binary` values) evaluated by tests/golden/make_wmse.py on the same batches."""
    z = np.load(os.path.join(GOLDEN, "kat_wmse.npz"))
    D = int(z["pred"].size)
    for binary in (False, True):
        eng = cls([D], int(z["H"]), int(z["O"]), batch_size=64, dropout_rate=0.0, learning_rate=0.0, seed=int(z["seed"]),
                  loss_binary=binary, **kw)
        eng.set_matrix(z["norm"])
        eng.set_indices(0, z["pred"], z["targ"])
        eng.gather(True)
        n = z["norm"].shape[0]
        eng.set_split(np.arange(n - 10, dtype=np.int32), np.arange(n - 10, n, dtype=np.int32))
        eng.init_weights()
        for i in range(2):
            rows = z["rows_%d" % i]
            np.testing.assert_allclose(eng.predict(rows), z["y_pred_%d" % i], rtol=1e-5, atol=1e-6)      # same y_pred as the fixture's
            loss = eng.train_step(rows, epoch_key=0, step_key=i)
            want = z["wmse_binary_%d" % i] if binary else z["wmse_%d" % i]
            np.testing.assert_allclose(loss[0], want, rtol=rtol, err_msg="wMSE(binary=%s) batch %d" % (binary, i))
            np.testing.assert_allclose(eng.predict(rows), z["y_pred_%d" % i], rtol=1e-5, atol=1e-6)      # lr = 0: nothing moved
        eng.close()


def check_reference_wmse_gradient(cls, rtol, atol, **kw):
    """dL/dy_hat of an optimiser step (lr = 0, no dropout) against central differences of the REFERENCE's own wMSE in float64
    (tests/golden/make_wmse.py, `dwmse_*`): the engine's dL/dz divided by sigmoid(z) (softplus' derivative)."""
    z = np.load(os.path.join(GOLDEN, "kat_wmse.npz"))
    D = int(z["pred"].size)
    for binary in (False, True):
        eng = cls([D], int(z["H"]), int(z["O"]), batch_size=64, dropout_rate=0.0, learning_rate=0.0, seed=int(z["seed"]), loss_binary=binary, **kw)
        eng.set_matrix(z["norm"])
        eng.set_indices(0, z["pred"], z["targ"])
        eng.gather(True)
        n = z["norm"].shape[0]
        eng.set_split(np.arange(n - 10, dtype=np.int32), np.arange(n - 10, n, dtype=np.int32))
        eng.init_weights()
        for i in range(2):
            rows = z["rows_%d" % i]
            eng.train_step(rows, epoch_key=0, step_key=i)
            dz, pre = eng.last_dz(0, rows.size)
            got = dz * (1.0 + np.exp(-pre))                      # / sigmoid(z)
            want = z["dwmse_binary_%d" % i] if binary else z["dwmse_%d" % i]
            assert want.shape == got.shape and np.abs(want).max() > 0
            np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg="d wMSE(binary=%s) / d y_hat, batch %d" % (binary, i))
        eng.close()


def check_against_keras_file(path, cls, require_real=True):
    """The consumer of tests/golden/make_keras.py's output: three train_on_batch steps (dropout rate 0) + predict of the kat_steps
    problem.  require_real: refuse a file that does not say which TensorFlow wrote it (the plumbing check feeds one made from
    the oracle itself, which pins nothing)."""
    ker, kat = np.load(path), load_kat()
    if require_real:
        assert "tf_version" in ker.files and "source" not in ker.files, "kat_keras.npz must come from real Keras (make_keras.py)"
    K, O = len(kat["Ds"]), int(kat["O"])
    eng = kat_engine(cls, kat, dropout_rate=0.0)       # Dropout(0): identity, no 1/(1-p) scale
    for t in range(3):
        loss = eng.train_step(kat["rows_%d" % t])
        np.testing.assert_allclose(loss, ker["loss"][t], rtol=2e-5)
    pred = eng.predict()
    for k in range(K):
        for got, name in zip(eng.get_weights(k), ("W1", "b1", "W2", "b2")):
            np.testing.assert_allclose(got, ker["%s_%d" % (name, k)], rtol=1e-4, atol=1e-6, err_msg="%s k=%d" % (name, k))
        np.testing.assert_allclose(pred[:, k * O:(k + 1) * O], ker["predict_%d" % k], rtol=1e-4, atol=1e-6)
    eng.close()
