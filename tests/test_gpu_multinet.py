"""GPU end-to-end tests of the drop-in surfaces: MultiNet.fit/predict and the deepImpute CLI on
the HIP engine (mirrors the reference's tests/multinet_test.py and tests/deepImpute_test.py,
which only check that the flow runs), plus what those tests do not assert: restore-policy
invariants, save/load round trip, agreement with the CPU oracle injected through the same shell."""
import argparse
import os
from unittest import mock

import numpy as np
import pandas as pd
import pytest

from helpers import multinet_with

pytestmark = pytest.mark.gpu


def _raw(n=300, g=700, seed=0):
    rng = np.random.default_rng(seed)
    u, v = rng.normal(size=(n, 6)), rng.normal(size=(g, 6))
    lam = np.exp(0.6 * (u @ v.T) / np.sqrt(6) + rng.normal(0.3, 0.8, size=g))      # planted low-rank structure
    counts = rng.poisson(lam).astype(np.float64)
    counts[:, :5] += rng.poisson(20, size=(n, 5))                                  # make sure max >= 10
    return pd.DataFrame(counts, index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])


def test_multinet_fit_predict_like_reference_test(tmp_path):
    from deepimpute_amd.multinet import MultiNet
    raw = _raw()
    net = MultiNet(architecture=[{"type": "dense", "activation": "relu", "neurons": 150},
                                 {"type": "dropout", "activation": "dropout", "rate": 0.2}],
                   loss="wMSE", sub_outputdim=128, seed=123, ncores=2, verbose=1, max_epochs=40,
                   learning_rate=1e-3, output_prefix=str(tmp_path))
    net.fit(raw)
    out = net.predict(raw, policy="restore")
    assert out.shape == raw.shape and list(out.columns) == list(raw.columns)
    assert np.isfinite(out.values).all()
    pos = raw.values > 0
    assert np.array_equal(out.values[pos], raw.values[pos])
    assert 1 <= net.trained_epochs <= 40 and len(net.history["val_loss"]) == net.trained_epochs
    assert net.history["val_loss"][-1] < net.history["val_loss"][0]               # it learns
    assert net.test_metrics["correlation"] > 0.3
    # a fresh object reloads the saved weights (predict() of the reference always reloads, :276)
    net2 = MultiNet(sub_outputdim=128, seed=123, ncores=2, output_prefix=str(tmp_path))
    net2.predictors, net2.targets = net.predictors, net.targets
    out2 = net2.predict(raw, policy="restore")
    np.testing.assert_allclose(out2.values, out.values, rtol=1e-6)
    only = net.predict(raw, imputed_only=True, policy="max")
    assert list(only.columns) == sorted(set(net.targets.flatten()))


def test_shell_on_hip_matches_shell_on_oracle(tmp_path):
    """Same MultiNet shell, same seed: HIP engine vs the CPU oracle injected as engine -> same
    early-stopping epoch and imputed values within 1e-4 relative (north-star tolerance)."""
    from deepimpute_amd.multinet import MultiNet
    from oracle.dimo import OracleEngine
    raw = _raw(n=220, g=400, seed=5)
    kw = dict(sub_outputdim=64, seed=7, ncores=1, verbose=0, max_epochs=5, patience=2, learning_rate=1e-3,
              architecture=[{"type": "dense", "neurons": 48, "activation": "relu"}, {"type": "dropout", "rate": 0.25}])
    a = MultiNet(output_prefix=str(tmp_path / "a"), **kw).fit(raw, NN_lim=128)
    b = multinet_with(OracleEngine, output_prefix=str(tmp_path / "b"), **kw).fit(raw, NN_lim=128)
    assert a.trained_epochs == b.trained_epochs
    np.testing.assert_allclose(a.history["val_loss"], b.history["val_loss"], rtol=2e-4)
    pa, pb = a.predict(raw, imputed_only=True), b.predict(raw, imputed_only=True)
    np.testing.assert_allclose(pa.values, pb.values, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(a.test_metrics["MSE"], b.test_metrics["MSE"], rtol=1e-3)


def test_shell_with_tanh_architecture_on_hip_matches_oracle(tmp_path):
    """A non-relu architecture through the whole shell (build -> engine activation -> save/load -> predict)."""
    from deepimpute_amd.multinet import MultiNet
    from oracle.dimo import OracleEngine
    raw = _raw(n=200, g=300, seed=9)
    kw = dict(sub_outputdim=64, seed=11, ncores=1, verbose=0, max_epochs=3, patience=3, learning_rate=1e-3,
              architecture=[{"type": "dense", "neurons": 40, "activation": "tanh"}, {"type": "dropout", "rate": 0.2}])
    a = MultiNet(output_prefix=str(tmp_path / "a"), **kw).fit(raw, NN_lim=128)
    b = multinet_with(OracleEngine, output_prefix=str(tmp_path / "b"), **kw).fit(raw, NN_lim=128)
    np.testing.assert_allclose(a.history["val_loss"], b.history["val_loss"], rtol=1e-4)
    np.testing.assert_allclose(a.predict(raw).values, b.predict(raw).values, rtol=1e-4, atol=1e-6)
    fresh = MultiNet(output_prefix=str(tmp_path / "a"), sub_outputdim=64, seed=11, ncores=1, verbose=0)   # reload: model.json carries the architecture
    fresh.predictors, fresh.targets = a.predictors, a.targets
    np.testing.assert_allclose(fresh.predict(raw).values, a.predict(raw).values, rtol=1e-6)


def test_cli_end_to_end(tmp_path):
    """The reference's deepImpute_test: parse_args mocked with a fixed Namespace, output None."""
    from deepimpute_amd.deepImpute import deepImpute
    raw = _raw(n=200, g=300, seed=9)
    path = str(tmp_path / "in.csv")
    raw.to_csv(path)
    args = dict(inputFile=path, cell_axis="rows", cores=1, learning_rate=1e-3, batch_size=64, max_epochs=6,
                output_neurons=64, hidden_neurons=300, dropout_rate=0.2, subset=1, limit=128, minVMR=0.5,
                n_pred=None, policy="restore", output=None)
    with mock.patch("argparse.ArgumentParser.parse_args", return_value=argparse.Namespace(**args)):
        out = deepImpute()
    assert out.shape == raw.shape
    # and through the file path with the string/float forms the real CLI produces
    args.update(limit="128", subset=150.0, output=str(tmp_path / "out.csv"))
    with mock.patch("argparse.ArgumentParser.parse_args", return_value=argparse.Namespace(**args)):
        assert deepImpute() is None
    back = pd.read_csv(args["output"], index_col=0)
    assert back.shape == raw.shape


def test_rccl_single_rank_comm_roundtrip():
    """World size 1 exercises the RCCL binding (dlopen, unique id, init, all-reduce, gather)."""
    from deepimpute_amd.engine import HipEngine
    e = HipEngine([20, 24], 32, 16, seed=3)
    rng = np.random.default_rng(0)
    e.set_matrix(np.log1p(rng.poisson(3.0, size=(70, 60))).astype(np.float32))
    for k in range(2):
        e.set_indices(k, rng.choice(60, [20, 24][k], replace=False), rng.choice(60, 16, replace=False))
    e.gather(True)
    e.init_weights()
    e.comm_init(e.comm_unique_id(), 1, 0)
    assert np.allclose(e.comm_allreduce_sum(np.array([1.5, 2.5])), [1.5, 2.5])
    ref = e.predict()
    e.predict_device()
    got = e.comm_gather_predictions(70, [2], root=0, is_root=True)
    assert np.array_equal(got, ref)
    e.comm_destroy()


def test_general_architecture_through_the_shell_matches_oracle(tmp_path):
    """What the tuned kernels do not take -- two hidden layers, batch 128, a keras loss by name -- through the whole
    shell on the general path (build -> HipGeneralEngine -> fit -> save/load -> predict) against the general oracle."""
    from deepimpute_amd.multinet import MultiNet
    from oracle.dimo import GeneralOracleEngine, OracleEngine

    class Oracles:
        def __new__(cls, *a, **k):
            return OracleEngine(*a, **k)
        general = staticmethod(GeneralOracleEngine)
    raw = _raw(n=400, g=300, seed=9)
    kw = dict(sub_outputdim=64, seed=11, ncores=1, verbose=0, max_epochs=3, patience=3, learning_rate=1e-3, batch_size=128, loss="mean_squared_error",
              architecture=[{"type": "dense", "neurons": 48, "activation": "relu"}, {"type": "dropout", "rate": 0.2},
                            {"type": "dense", "neurons": 32, "activation": "tanh"}, {"type": "dropout", "rate": 0.1}])
    a = MultiNet(output_prefix=str(tmp_path / "a"), **kw).fit(raw, NN_lim=128)
    b = multinet_with(Oracles, output_prefix=str(tmp_path / "b"), **kw).fit(raw, NN_lim=128)
    assert type(a._engine).__name__ == "HipGeneralEngine" and a.trained_epochs == b.trained_epochs == 3
    np.testing.assert_allclose(a.history["val_loss"], b.history["val_loss"], rtol=1e-4)
    np.testing.assert_allclose(a.predict(raw).values, b.predict(raw).values, rtol=1e-4, atol=1e-6)
    fresh = MultiNet(output_prefix=str(tmp_path / "a"), sub_outputdim=64, seed=11, ncores=1, verbose=0)   # model.json carries architecture, loss, batch size
    fresh.predictors, fresh.targets = a.predictors, a.targets
    np.testing.assert_allclose(fresh.predict(raw).values, a.predict(raw).values, rtol=1e-6)
    assert type(fresh._engine).__name__ == "HipGeneralEngine"


def test_cli_with_batch_128_and_hidden_512(tmp_path):
    """`deepImpute --batch-size 128 --hidden-neurons 512` (legal for the reference, parser.py:50-66) runs on the general path."""
    from deepimpute_amd.deepImpute import deepImpute
    raw = _raw(n=300, g=300, seed=9)
    path = str(tmp_path / "in.csv")
    raw.to_csv(path)
    args = dict(inputFile=path, cell_axis="rows", cores=1, learning_rate=1e-3, batch_size=128, max_epochs=4, output_neurons=64, hidden_neurons=512,
                dropout_rate=0.2, subset=1, limit=128, minVMR=0.5, n_pred=None, policy="restore", output=None)
    with mock.patch("argparse.ArgumentParser.parse_args", return_value=argparse.Namespace(**args)):
        out = deepImpute()
    assert out.shape == raw.shape and np.isfinite(out.values).all()
    pos = raw.values > 0
    assert np.array_equal(out.values[pos], raw.values[pos])


def test_resident_counts_path_equals_host_path(tmp_path, monkeypatch):
    """fit() / predict() with the raw counts uploaded ONCE and read in place on the GPU (correlation, log1p through numpy's
    table, restore against the observed counts) must give, bit for bit, what the host path gives -- same predictor lists, same
    training history, same imputed frame -- for the frame that was fitted (found resident again by its checksum), for another
    frame (uploaded afresh), for a frame edited in place after fit (the checksum notices), and it must step aside for data
    that are not counts."""
    from deepimpute_amd.multinet import MultiNet
    raw = _raw(n=260, g=520, seed=3)
    kw = dict(sub_outputdim=64, seed=11, ncores=1, verbose=0, max_epochs=4, patience=10, learning_rate=1e-3)
    runs = {}
    for mode in ("1", "1h", "0"):                                       # "1h": resident counts, statistics by the host routines
        monkeypatch.setenv("DIMN_RESIDENT_COUNTS", mode[0])
        monkeypatch.setenv("DIMN_DEVICE_STATS", "0" if mode == "1h" else "1")
        net = MultiNet(output_prefix=str(tmp_path / mode), **kw).fit(raw, NN_lim=200)
        assert (getattr(net, "_resident", None) is not None) == (mode[0] == "1")
        same = net.predict(raw)
        used_resident = "predict.log1p" not in net.timings
        assert used_resident == (mode[0] == "1")
        other = raw.iloc[::-1].copy()                                   # another frame: same genes, cells in reverse order
        flipped = net.predict(other, policy="max")
        edited = raw.copy()
        edited.iloc[5, 7] += 3.0
        after_edit = net.predict(edited)
        runs[mode] = (net.predictors, net.history, same, flipped, after_edit, net.test_metrics)
        net.close()
    for other_mode in ("0", "1h"):
        a, b = runs["1"], runs[other_mode]
        assert len(a[0]) == len(b[0]) and all(list(x) == list(y) for x, y in zip(a[0], b[0]))
        assert a[1] == b[1]
        for i in (2, 3, 4):
            assert a[i].index.equals(b[i].index) and np.array_equal(a[i].values, b[i].values), i
        assert a[5] == b[5]
    a = runs["1"]
    assert a[4].iloc[5, 7] == raw.iloc[5, 7] + 3.0                     # the edited count was restored, not the stale resident one
    # not counts: the upload declines, the host path runs
    monkeypatch.setenv("DIMN_RESIDENT_COUNTS", "1")
    scaled = raw * 1.5
    net = MultiNet(output_prefix=str(tmp_path / "s"), **kw).fit(scaled, NN_lim=200)
    assert getattr(net, "_resident", None) is None and np.isfinite(net.predict(scaled).values).all()
    net.close()
