"""Pins deepimpute_amd.multinet.MultiNet's host shell against fixtures captured from the imported
REFERENCE (tests/golden/make_shell.py): same genes, targets, predictors, validation split and
predict() post-processing for the same seed and data.  The network itself is replaced on both
sides by the same deterministic fake prediction, so these tests need no GPU."""
import json
import os
import sys

import numpy as np
import pandas as pd
import pytest

from helpers import GOLDEN, multinet_with
sys.path.insert(0, GOLDEN)
from make_shell import fake_prediction  # noqa: E402  (pure function shared with the capture script)

from deepimpute_amd.multinet import MultiNet, get_distance_matrix, inspect_data, wMSE  # noqa: E402


class FakeEngine:
    """Records what the shell hands to the engine; predict() = the capture script's stand-in."""
    instances = []

    def __init__(self, D, hidden, out_dim, **kw):
        self.D, self.K, self.H, self.O, self.kw = list(D), len(D), hidden, out_dim, kw
        self.pred, self.targ = {}, {}
        FakeEngine.instances.append(self)

    def set_matrix(self, norm):
        assert norm.dtype == np.float32
        self.norm = np.array(norm)

    def set_indices(self, k, p, t):
        self.pred[k], self.targ[k] = np.array(p), np.array(t)

    def gather(self, with_targets=True):
        self.gathered = with_targets

    def set_split(self, train, val):
        self.train, self.val = np.array(train), np.array(val)

    def init_weights(self, seed=None):
        self.init_seed = seed

    def fit(self, max_epochs, patience):
        self.fit_args = (max_epochs, patience)
        return 3, np.array([3.0, 2.0, 1.0]), np.array([3.0, 2.0, 1.0])

    def predict(self, rows=None):
        x = self.norm if rows is None else self.norm[np.asarray(rows)]
        return np.hstack([fake_prediction(x[:, self.pred[k]], k, self.O) for k in range(self.K)])

    def get_weights(self, k):
        z = lambda *s: np.zeros(s, np.float32)
        return z(self.D[k], self.H), z(self.H), z(self.H, self.O), z(self.O)

    def close(self):
        pass


CASES = json.load(open(os.path.join(GOLDEN, "shell_cases.json")))
ARR = np.load(os.path.join(GOLDEN, "shell_cases.npz"))


def _raw(name):
    v = ARR[name + "/raw"].astype(np.float64)
    return pd.DataFrame(v, index=["c%d" % i for i in range(v.shape[0])], columns=["g%d" % j for j in range(v.shape[1])])


@pytest.mark.parametrize("name", sorted(CASES))
def test_shell_matches_reference_capture(name, tmp_path, capsys):
    meta = CASES[name]
    raw = _raw(name)
    fitkw = dict(meta["fit"])
    net = multinet_with(FakeEngine, output_prefix=str(tmp_path), **meta["ctor"])
    net.fit(raw, **fitkw)
    eng = FakeEngine.instances[-1]
    col = {c: i for i, c in enumerate(raw.columns)}

    # plan: targets / predictors identical (labels and order)
    targets = np.array([[col[x] for x in t] for t in net.targets], np.int32)
    assert np.array_equal(targets, ARR[name + "/targets"])
    assert len(net.predictors) == meta["K"]
    for k in range(meta["K"]):
        assert np.array_equal(np.array([col[x] for x in net.predictors[k]], np.int32), ARR["%s/pred%d" % (name, k)])
        assert np.array_equal(eng.targ[k], ARR[name + "/targets"][k])

    # validation split: same cells; train rows label-sorted as np.setdiff1d gives them
    used = ARR[name + "/used_rows"]
    assert np.array_equal(used[eng.val], ARR[name + "/test_cells"])
    assert np.array_equal(used[eng.train], ARR[name + "/train_cells"])

    # what reaches the engine == what the reference handed to Keras
    layers = meta["layers"]
    drop = [l for l in layers if l[0] == "Dropout"]          # K Dense(H), K Dropout, K Dense(O)
    assert len(layers) == 3 * meta["K"] and len(drop) == meta["K"]
    assert eng.H == int(layers[0][1][0]) and eng.O == int(layers[-1][1][0])
    assert "relu" in layers[0][2]["activation"] and "softplus" in layers[-1][2]["activation"]
    assert abs(eng.kw["dropout_rate"] - float(drop[0][1][0])) < 1e-12
    assert int(drop[0][2]["seed"]) == meta["ctor"]["seed"] == eng.kw["seed"]
    assert eng.kw["learning_rate"] == 1e-4 and eng.kw["batch_size"] == meta["fit_call"]["batch_size"]
    assert eng.fit_args == (meta["fit_call"]["epochs"], 5)
    assert eng.D == [s[1] for s in meta["fit_call"]["x_shapes"]]
    assert net.trained_epochs == meta["trained_epochs"] == 3
    assert os.path.exists(os.path.join(str(tmp_path), "model.json"))

    # held-out metrics on the fake prediction
    np.testing.assert_allclose([net.test_metrics["correlation"], net.test_metrics["MSE"]],
                               ARR[name + "/test_corr_mse"], rtol=1e-5)

    # predict(): post-processing identical for both policies and imputed_only
    for policy in ("restore", "max"):
        got = net.predict(raw, policy=policy)
        assert list(got.columns) == list(raw.columns) and list(got.index) == list(raw.index)
        np.testing.assert_allclose(got.values, ARR["%s/imputed_%s" % (name, policy)], rtol=1e-6, atol=1e-7)
    only = net.predict(raw, imputed_only=True)
    assert np.array_equal(np.array([col[x] for x in only.columns], np.int32), ARR[name + "/imputed_only_cols"])
    np.testing.assert_allclose(only.values, ARR[name + "/imputed_only"], rtol=1e-6, atol=1e-7)
    out = capsys.readouterr().out
    for msg in ("genes selected for imputation", "Net 0:", "Normalization", "Building network",
                "Fitting with", "Stopped fitting after 3 epochs", "Saved model to disk in"):
        if name == "gene_list" and msg.startswith("genes selected"):
            continue
        assert msg in out


def test_restore_policy_invariants():
    raw = _raw("default64")
    net = multinet_with(FakeEngine, seed=123, sub_outputdim=64, ncores=1, verbose=0)
    net.fit(raw)
    out = net.predict(raw, policy="restore")
    pos = raw.values > 0
    assert np.array_equal(out.values[pos], raw.values[pos])          # observed counts are returned exactly
    targets = set(net.targets.flatten())
    non_target = [c for c in raw.columns if c not in targets]
    assert np.all(out[non_target].values[raw[non_target].values == 0] == 0)   # untouched zeros stay zero


def test_module_functions():
    raw = _raw("progressive")
    d = get_distance_matrix(raw)
    assert d.shape[0] == d.shape[1] and np.allclose(np.diag(d.values), 1.0)
    assert (d.values >= 0).all()
    assert get_distance_matrix(raw, n_pred=50).shape == (50, 50)
    y = np.array([[0.0, 2.0], [1.0, 0.0]]); yh = np.array([[5.0, 1.0], [0.0, 7.0]])
    assert np.isclose(wMSE(y, yh), (0 + 2 * 1 + 1 * 1 + 0) / 4)
    assert np.isclose(wMSE(y, yh, binary=True), (0 + 1 + 1 + 0) / 4)
    with pytest.raises(SystemExit):
        inspect_data(np.log1p(raw) * 0 + 1.0)          # max < 10 -> looks log-transformed
    dup = raw.copy(); dup.index = ["c0"] * len(dup)
    with pytest.raises(SystemExit):
        inspect_data(dup)


def test_unsupported_architecture_is_loud():
    net = multinet_with(FakeEngine, ncores=1,
                   architecture=[{"type": "dense", "neurons": 8, "activation": "softmax"}])
    with pytest.raises(NotImplementedError):
        net.build([10])
    with pytest.raises(NotImplementedError):           # two hidden layers: the general path, which this injected factory lacks
        multinet_with(FakeEngine, ncores=1, architecture=[{"type": "dense", "neurons": 8, "activation": "relu"},
                                                                    {"type": "dense", "neurons": 8, "activation": "relu"}]).build([10])
    with pytest.raises(NotImplementedError):           # dropout on the inputs: the general path, which this injected factory lacks
        multinet_with(FakeEngine, ncores=1, architecture=[{"type": "dropout", "rate": 0.1},
                                                                    {"type": "dense", "neurons": 8, "activation": "relu"}]).build([10])
    eng = multinet_with(FakeEngine, ncores=1,
                   architecture=[{"type": "dense", "neurons": 8, "activation": "tanh"}, {"type": "dropout", "rate": 0.1}]).build([10])
    assert eng.kw["activation"] == "tanh" and eng.H == 8 and abs(eng.kw["dropout_rate"] - 0.1) < 1e-12
    with pytest.raises(SystemExit):                    # the reference's "Unknown loss ... Aborting." (multinet.py:160-161)
        multinet_with(FakeEngine, ncores=1, loss="not_a_loss").build([10])


def test_general_architectures_reach_the_general_engine():
    """build() routes what the tuned kernels do not take -- several hidden layers, hidden > 384, batch > 64, keras losses
    by name -- to the general constructor with the parsed layer list (multinet.py:135-162, parser.py:50-66)."""
    calls = []

    class Factory(FakeEngine):
        @staticmethod
        def general(D, layers, out_dim, **kw):
            calls.append((list(D), list(layers), out_dim, kw))
            return "general"
    arch = [{"type": "dense", "neurons": 600, "activation": "relu"}, {"type": "dropout", "rate": 0.3},
            {"type": "dense", "neurons": 64, "activation": "tanh"}]
    assert multinet_with(Factory, ncores=1, architecture=arch, batch_size=128, loss="mean_squared_error", seed=7).build([10, 12]) == "general"
    D, layers, out_dim, kw = calls[-1]
    assert D == [10, 12] and layers == [(600, "relu", 0.3), (64, "tanh", 0.0)] and out_dim == 512
    assert kw["batch_size"] == 128 and kw["loss"] == "mean_squared_error" and kw["seed"] == 7
    # each of the three limits of the tuned kernels alone is enough
    for extra in (dict(batch_size=65), dict(architecture=[{"type": "dense", "neurons": 400, "activation": "relu"}]), dict(loss="mae")):
        assert multinet_with(Factory, ncores=1, **extra).build([10]) == "general"
    # and the default family stays on the tuned kernels
    eng = multinet_with(Factory, ncores=1).build([10])
    assert isinstance(eng, FakeEngine) and eng.H == 256


def test_cli_parser_matches_reference_flags(monkeypatch):
    from deepimpute_amd.parser import build_parser
    a = build_parser().parse_args(["in.csv"])
    assert (a.output, a.cores, a.cell_axis, a.limit, a.minVMR, a.subset) == ("./imputed.csv", -1, "rows", "auto", 0.5, 1)
    assert (a.learning_rate, a.batch_size, a.max_epochs, a.hidden_neurons) == (0.0005, 64, 300, 300)
    assert (a.dropout_rate, a.output_neurons, a.n_pred, a.policy) == (0.2, 512, None, "restore")
    b = build_parser().parse_args(["in.csv", "--limit", "2000", "--subset", "200", "--cell-axis", "columns", "-o", "x.csv"])
    assert b.limit == "2000" and b.subset == 200.0 and b.cell_axis == "columns" and b.output == "x.csv"


def test_cli_digit_limit_and_subset_are_coerced(tmp_path, monkeypatch):
    """`--limit 64` arrives as a str and `--subset 90` as a float (parser.py:26,38); the reference
    crashes on both, the drop-in coerces them."""
    raw = _raw("progressive")
    net = multinet_with(FakeEngine, seed=99, sub_outputdim=32, ncores=1, verbose=0, output_prefix=str(tmp_path))
    net.fit(raw, NN_lim="64", cell_subset=90.0)
    assert net.targets.shape[1] == 32 and FakeEngine.instances[-1].norm.shape[0] == 90


def test_distance_matrix_backends_without_gpu():
    """No GPU here: 'auto' falls back to the reference's numpy computation, 'hip' is loud."""
    raw = _raw("progressive")
    a = get_distance_matrix(raw, backend="auto")
    b = get_distance_matrix(raw, backend="numpy")
    assert np.array_equal(a.values, b.values)
    import subprocess, sys
    code = ("import numpy as np\n"
            "from deepimpute_amd.multinet import _abs_corrcoef\n"
            "try:\n"
            "    _abs_corrcoef(np.random.rand(20, 5), backend='hip'); print('RAN')\n"
            "except Exception as e:\n"
            "    print('LOUD', type(e).__name__)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout
    assert "LOUD" in out, out


def test_host_pool_blocks_are_bit_identical(monkeypatch):
    """The host pool runs the reference's own pandas/numpy calls on column / row blocks: every value must
    be what the whole-matrix call gives, to the bit (so the plan -- gene order, predictors -- is the same)."""
    from deepimpute_amd import _hostpar
    monkeypatch.setenv("DIMN_HOST_THREADS", "4")
    monkeypatch.setattr(_hostpar, "_block_for", lambda *a, **k: 16)
    rng = np.random.default_rng(5)
    v = rng.poisson(rng.gamma(0.4, 4.0, size=(1, 75)), size=(301, 75)).astype(np.float64)
    v[:, 3] = 0.0                                          # a constant gene: std/mean = nan
    for dtype in (np.float64, np.float32, np.int64):
        raw = pd.DataFrame(v.astype(dtype), index=["c%d" % i for i in range(301)], columns=["g%d" % j for j in range(75)])
        var, mean = _hostpar.column_var_mean(raw)
        assert var.index.equals(raw.var().index)
        assert np.array_equal(var.values, raw.var().values, equal_nan=True)
        assert np.array_equal(mean.values, raw.mean().values, equal_nan=True)
        assert np.array_equal(np.sqrt(var).values, raw.std().values, equal_nan=True)
        got, want = _hostpar.log1p_float32(raw), np.log1p(raw).astype(np.float32)
        assert got.values.dtype == np.float32 and np.array_equal(got.values, want.values)
        assert got.index.equals(raw.index) and got.columns.equals(raw.columns)
    pos = rng.permutation(75)[:40]
    assert np.array_equal(_hostpar.take_columns(v, pos), v[:, pos])
    sq = rng.normal(size=(70, 70)); sq[rng.random((70, 70)) < 0.1] = np.nan
    assert np.array_equal(_hostpar.zero_nans_inplace(sq.copy()), pd.DataFrame(sq).fillna(0).values)


def test_predict_row_blocks_do_not_change_the_result(monkeypatch):
    from deepimpute_amd import multinet as mn
    raw = _raw("default64")
    net = multinet_with(FakeEngine, seed=123, sub_outputdim=64, ncores=1, verbose=0)
    net.fit(raw)
    whole = {p: net.predict(raw, policy=p).values for p in ("restore", "max", None)}
    monkeypatch.setattr(mn, "_POST_ROWS", 7)
    monkeypatch.setenv("DIMN_HOST_THREADS", "3")
    for p, want in whole.items():
        assert np.array_equal(net.predict(raw, policy=p).values, want, equal_nan=True)


@pytest.mark.parametrize("kind", ["counts", "real"])
def test_native_gene_statistics_are_pandas_to_the_bit(kind):
    """fit() orders the genes by raw.var() / (1 + raw.mean()) (reference multinet.py:191): libdimn's threaded host routine
    (dimn_col_stats) must give pandas' numbers to the bit -- mean as a sequential sum over the rows, var as nanvar computes it
    (its own pairwise-summed average, pairwise-summed squared deviations, numpy's 8192-element chunks) -- for row counts on
    both sides of every block boundary of numpy's pairwise summation; a NaN sends the call back to pandas."""
    from deepimpute_amd import _hostpar
    rng = np.random.default_rng(5)
    for n, g in ((2, 5), (7, 3), (8, 3), (129, 70), (1024, 64), (5000, 130), (8192, 9), (8193, 64), (20011, 33)):
        a = rng.poisson(rng.gamma(2.0, 1.5, size=(n, g))).astype(np.float64) if kind == "counts" else rng.normal(size=(n, g)) * 1e3 + 5
        raw = pd.DataFrame(a, columns=["g%d" % j for j in range(g)])
        assert _hostpar._native_col_stats(raw.values) is not None            # the native routine is what runs
        var, mean = _hostpar.column_var_mean(raw)
        assert np.array_equal(var.values, raw.var().values), (n, g)
        assert np.array_equal(mean.values, raw.mean().values), (n, g)
        assert var.index.equals(raw.columns) and mean.index.equals(raw.columns)
        assert _hostpar.matrix_max(raw.values) == raw.values.max()
    raw.iloc[3, 2] = np.nan
    assert _hostpar._native_col_stats(raw.values) is None
    var, mean = _hostpar.column_var_mean(raw)
    assert np.array_equal(var.values, raw.var().values) and np.array_equal(mean.values, raw.mean().values)
