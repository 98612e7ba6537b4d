"""GPU tests of the planning rows of SURVEY 8f that run on the device: setPredictors fused behind the correlation
(dimn_select_predictors) -- against the predictor lists captured from the imported REFERENCE (tests/golden/shell_cases.*)
and against the host implementation on a 5k-gene synthetic."""
import numpy as np
import pandas as pd
import pytest

from helpers import multinet_with

import test_shell as shell                     # FakeEngine, fixtures (CPU module; importing it runs nothing)
from deepimpute_amd.multinet import MultiNet, get_distance_matrix

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(shell.CASES))
def test_device_predictor_selection_matches_reference_capture(name, tmp_path, monkeypatch):
    meta = shell.CASES[name]
    raw = shell._raw(name)
    used = []
    real = MultiNet._set_predictors_device
    monkeypatch.setattr(MultiNet, "_set_predictors_device", lambda self, *a, **k: used.append(real(self, *a, **k)) or used[-1])
    net = multinet_with(shell.FakeEngine, output_prefix=str(tmp_path), **meta["ctor"])
    net.fit(raw, **dict(meta["fit"]))
    assert used == [True], "the fused device selection did not run"
    col = {c: i for i, c in enumerate(raw.columns)}
    assert len(net.predictors) == meta["K"]
    for k in range(meta["K"]):
        assert np.array_equal(np.array([col[x] for x in net.predictors[k]], np.int32), shell.ARR["%s/pred%d" % (name, k)]), k


@pytest.mark.parametrize("ntop", [5, 7, 16])
def test_device_predictor_selection_matches_host_on_5k_genes(ntop, tmp_path):
    rng = np.random.default_rng(5)
    n, g = 600, 5000
    u, v = rng.normal(size=(n, 8)), rng.normal(size=(g, 8))
    lam = np.exp(0.7 * (u @ v.T) / np.sqrt(8) + rng.normal(0.2, 0.9, size=g))
    counts = rng.poisson(lam).astype(np.float64)
    counts[:, :3] += 12
    counts[:, 100] = 0                                                     # a constant gene: never a candidate
    raw = pd.DataFrame(counts, index=["c%d" % i for i in range(n)], columns=["g%05d" % j for j in rng.permutation(g)])
    net = MultiNet(output_prefix=str(tmp_path), sub_outputdim=512, seed=3, verbose=0)
    np.random.seed(3)
    var, mean = raw.var(), raw.mean()
    metric = (var / (1 + mean)).sort_values(ascending=False)
    genes = net.filter_genes(metric[metric > 0], 0.5, NN_lim=4000)
    net.setTargets(pd.DataFrame(columns=pd.Index(genes)))
    assert net._set_predictors_device(raw, None, ntop, (var, mean))
    dev = [list(p) for p in net.predictors]
    net.setPredictors(get_distance_matrix(raw, backend="hip"), ntop=ntop)       # host selection over the same GPU correlation
    host = [list(p) for p in net.predictors]
    assert len(dev) == len(host) == 9
    for k, (a, b) in enumerate(zip(dev, host)):
        assert a == b, "sub-net %d: first difference at %d" % (k, next(i for i, (x, y) in enumerate(zip(a, b)) if x != y))


@pytest.mark.parametrize("name", sorted(shell.CASES))
def test_device_postprocessing_matches_host_path(name, tmp_path, monkeypatch):
    """predict()'s post-processing as the device epilogue (dimn_impute_finish) against the host implementation (which the
    reference-captured fixtures pin, tests/test_shell.py) on the five shell cases, both policies, no policy and
    imputed_only: identical wherever the observed counts win; elsewhere expm1/log1p run in float64 on the device
    (ocml) instead of glibc -- both within an ulp of the true value, so at most 2 ulp apart."""
    meta = shell.CASES[name]
    raw = shell._raw(name)
    ctor = dict(meta["ctor"])
    ctor.update(max_epochs=3, verbose=0)
    net = MultiNet(output_prefix=str(tmp_path), **ctor)
    net.fit(raw, **dict(meta["fit"]))
    used = []
    real = MultiNet._finish_on_device

    def spy(self, *a, **k):
        used.append(real(self, *a, **k))
        return used[-1]
    for policy in ("restore", "max", None):
        monkeypatch.setattr(MultiNet, "_finish_on_device", spy)
        dev = net.predict(raw, policy=policy)
        assert used and used[-1] is not None and used[-1] is not False
        monkeypatch.setattr(MultiNet, "_finish_on_device", lambda self, *a, **k: None)
        host = net.predict(raw, policy=policy)
        assert list(dev.columns) == list(host.columns) and list(dev.index) == list(host.index)
        a, b = dev.values, host.values
        assert a.dtype == b.dtype == np.float64 and np.isfinite(a).all()
        np.testing.assert_allclose(a, b, rtol=4.5e-16, atol=0)
        if policy == "restore":
            seen = raw.values > 0
            assert np.array_equal(a[seen], raw.values[seen])
        assert (a == b).mean() > 0.5        # the rest: 1-2 ulp (ocml vs glibc expm1)
    monkeypatch.setattr(MultiNet, "_finish_on_device", spy)
    only = net.predict(raw, imputed_only=True)
    assert list(only.columns) == sorted(set(net.targets.flatten()))


def test_device_held_out_metrics_match_scipy_path(tmp_path):
    """fit()'s test_metrics from the seven device sums (dimn_val_metrics, float64) against the reference's own
    scipy.stats.pearsonr / numpy MSE over the same predictions (multinet.py:251-262, float32 arithmetic)."""
    raw = shell._raw("default64")
    net = MultiNet(output_prefix=str(tmp_path), sub_outputdim=64, seed=11, verbose=0, max_epochs=4, learning_rate=1e-3,
                   architecture=[{"type": "dense", "neurons": 48, "activation": "relu"}, {"type": "dropout", "rate": 0.2}])
    net.fit(raw, NN_lim=128)
    dev = net.test_metrics

    class HostOnly:                                   # the same engine without the device metrics: the scipy path runs
        def __init__(self, e):
            self._e = e

        def predict(self, rows=None):
            return self._e.predict(rows)
    np.random.seed(11)
    norm = np.log1p(raw).astype(np.float32)
    held = np.random.choice(norm.index, int(0.05 * norm.shape[0]), replace=False)
    host = net._held_out_metrics(HostOnly(net._engine), norm, held, norm.index.get_indexer(held))
    assert abs(float(dev["correlation"]) - float(host["correlation"])) < 2e-5
    np.testing.assert_allclose(float(dev["MSE"]), float(host["MSE"]), rtol=2e-5)


def test_streamed_correlation_matches_resident(monkeypatch):
    """|corr| with the counts streamed in row blocks (two passes: column sums, then centre + accumulate on the fp64 matrix
    cores) -- what a matrix too large for the GPU takes (configs[4]) -- against the resident single pass and numpy."""
    from deepimpute_amd.multinet import _abs_corrcoef
    rng = np.random.default_rng(2)
    x = rng.poisson(rng.gamma(0.8, 3.0, size=700), size=(1111, 700)).astype(np.float64)
    x[:, 17] = 3.0                                                     # a constant gene: NaN -> 0
    resident = _abs_corrcoef(x, backend="hip")
    monkeypatch.setenv("DIMN_CORR_BUDGET_GB", "0:208")                 # force the streamed form, in six row blocks of 208, the last one partial
    streamed = _abs_corrcoef(x, backend="hip")
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = np.nan_to_num(np.abs(np.corrcoef(x.T)), nan=0.0)
    np.testing.assert_allclose(streamed, resident, rtol=0, atol=1e-13)
    np.testing.assert_allclose(streamed, ref, rtol=0, atol=1e-12)
    assert streamed[17].max() == 0.0


@pytest.mark.parametrize("n,g,O,K", [(300, 700, 64, 3), (9000, 1531, 32, 5), (8192 + 77, 640, 128, 2), (70, 5, 16, 1), (260, 900, 512, 76)])
# (70 x 5: a wave's eighth of a row is empty; 76 x 512 = 38 912 slots: the prediction row no longer fits the kernel's LDS stage)
def test_restore_epilogue_sends_only_the_zeros_and_equals_the_dense_epilogue(n, g, O, K):
    """dimn_impute_finish_restore (policy "restore" over resident counts: only the zero entries cross PCIe, the host merges them into
    a copy of its own frame) against dimn_impute_finish(raw = NULL, policy = 1) on the same prediction: bit for bit -- ragged column
    counts (a wave's eighth of a row is not a multiple of 64), more rows than workgroups, rows without zeros and all-zero rows, genes in
    several slots and genes in none, the chunked forward (>= 8192 rows) -- and a frame that is NOT the resident matrix is refused (the
    engine falls back to the dense epilogue)."""
    from deepimpute_amd._counts import DeviceCounts
    from deepimpute_amd.engine import HipEngine
    rng = np.random.default_rng(n + g)
    raw = rng.poisson(rng.gamma(0.6, 2.0, size=g), size=(n, g)).astype(np.float64)
    raw[3] = 0.0                                              # an all-zero cell
    raw[5] = np.maximum(raw[5], 1.0)                          # a cell without zeros
    raw[:, :4] += rng.poisson(12, size=(n, 4))
    D = [min(g - 1, 40 + 3 * (k % 9)) for k in range(K)]
    eng = HipEngine(D, 32, O, seed=5)
    counts = DeviceCounts.try_create(raw, 0)
    assert counts is not None
    pool = rng.permutation(g)
    if K * O - 7 <= g:
        slots = np.concatenate([pool[: K * O - 7], pool[:7]])     # seven genes occupy two slots; the genes past K * O - 7 none
    else:
        slots = np.concatenate([pool[:g - 3], rng.choice(pool[:g - 3], K * O - (g - 3))])   # many slots per gene, three genes in none
    for k in range(K):
        eng.set_indices(k, rng.choice(g, D[k], replace=False), slots[k * O:(k + 1) * O])
    eng.set_matrix_counts(counts)
    eng.gather(False)
    eng.init_weights()
    order = np.lexsort((np.arange(len(slots)), slots))
    gene_off = np.zeros(g + 1, np.int64)
    np.cumsum(np.bincount(slots, minlength=g), out=gene_off[1:])
    ceiling = 2 * np.log1p(raw.max())
    eng.predict_device()
    dense = eng.impute_finish(None, gene_off, order, "restore", ceiling)
    eng.predict_device()
    packed = eng.impute_finish(None, gene_off, order, "restore", ceiling, observed=raw)
    assert np.array_equal(dense, packed)
    assert eng.last_observed_checksum == counts.checksum              # the merge read every element: it knows the frame is the uploaded one
    seen = raw > 0
    assert np.array_equal(packed[seen], raw[seen]) and (packed[~seen] >= 0).all() and (packed[3] > 0).any()
    # the C entry point itself refuses a frame with other zeros ...
    other = raw.copy()
    c = min(11, g - 1)
    other[7, c] = 0.0 if raw[7, c] > 0 else 4.0
    from deepimpute_amd import _cabi
    out = np.empty_like(raw)
    rc = eng._f["impute_finish_restore"](eng._h, other.ctypes.data, 0, n, g, _cabi.p_i32(np.ascontiguousarray(gene_off, np.int32)),
                                         _cabi.p_i32(np.ascontiguousarray(order, np.int32)), float(ceiling), 0, _cabi.p_f64(out), None)
    assert rc == -3
    # ... and the engine says so instead of finishing the RESIDENT counts behind the caller's back (a dense epilogue the caller throws
    # away); MultiNet.predict() then uploads the frame it was given and runs the ordinary sequence once
    from deepimpute_amd.engine import FrameMismatch
    with pytest.raises(FrameMismatch):
        eng.impute_finish(None, gene_off, order, "restore", ceiling, observed=other)
    # an int64 frame of the same counts (what pd.read_csv hands over) is merged in place: same float64 result, same checksum
    eng.predict_device()
    assert np.array_equal(eng.impute_finish(None, gene_off, order, "restore", ceiling, observed=raw.astype(np.int64)), dense)
    assert eng.last_observed_checksum == counts.checksum
    eng.close(); counts.close()
