"""GPU tests of the planning rows of SURVEY 8f that run on the device: setPredictors fused behind the correlation
(dimn_select_predictors) -- against the predictor lists captured from the imported REFERENCE (tests/golden/shell_cases.*)
and against the host implementation on a 5k-gene synthetic."""
import numpy as np
import pandas as pd
import pytest

import test_shell as shell                     # FakeEngine, fixtures (CPU module; importing it runs nothing)
from deepimpute_amd.multinet import MultiNet, get_distance_matrix

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(shell.CASES))
def test_device_predictor_selection_matches_reference_capture(name, tmp_path, monkeypatch):
    meta = shell.CASES[name]
    raw = shell._raw(name)
    used = []
    real = MultiNet._set_predictors_device
    monkeypatch.setattr(MultiNet, "_set_predictors_device", lambda self, *a: used.append(real(self, *a)) or used[-1])
    net = MultiNet(output_prefix=str(tmp_path), engine_factory=shell.FakeEngine, **meta["ctor"])
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
