"""GPU parity of the GENERAL path (dimn_create_general: any architecture / batch size / loss; reference
deepimpute/multinet.py:135-162, parser.py:50-66) against torch-fp64 goldens and the general CPU oracle."""
import numpy as np
import pytest

from helpers import check_general_kat, make_problem

pytestmark = pytest.mark.gpu


def _hip():
    from deepimpute_amd.engine import HipGeneralEngine
    return HipGeneralEngine


def _oracle():
    from oracle.dimo import GeneralOracleEngine
    return GeneralOracleEngine


@pytest.mark.parametrize("loss", ["wmse", "wmse_binary", "mse", "mae", "msle", "logcosh", "huber", "poisson", "wmse+input_dropout"])
def test_general_kat_matches_autograd_golden(loss):
    check_general_kat(_hip(), loss, rtol=3e-4, atol=2e-6)


def _pair(prob, layers, classes=None, **kw):
    out = []
    for cls in (classes or (_hip(), _oracle())):
        e = cls(prob["Ds"], layers, prob["O"], **kw)
        e.set_matrix(prob["norm"])
        for k in range(len(prob["Ds"])):
            e.set_indices(k, prob["pred"][k], prob["targ"][k])
        e.gather(True)
        e.set_split(prob["train"], prob["val"])
        e.init_weights()
        out.append(e)
    return out


@pytest.mark.parametrize("layers,B,loss", [
    ([(512, "relu", 0.2)], 128, "wmse"),                                  # hidden > 384, batch > 64 (deepImpute --batch-size 128)
    ([(96, "relu", 0.3), (64, "tanh", 0.0), (48, "elu", 0.1)], 200, "wmse"),   # three hidden layers, one without dropout
    ([(300, "relu", 0.2)], 333, "mse"),                                   # odd batch, keras loss by name
    ([(40, "sigmoid", 0.0)], 16, "mae"),
    ([(0, "linear", 0.2), (96, "gelu", 0.3), (64, "selu", 0.0)], 100, "huber"),   # a Dropout layer before the first Dense layer; round-5 activations and loss
    ([(50, "relu", 0.1), (30, "tanh", 0.0)], 64, "wmse"),                 # widths that are not multiples of 4: the batch-row GEMMs stay on the LDS-staged k_gen_gemm
])
def test_general_two_epochs_match_oracle(layers, B, loss):
    prob = make_problem(n=700, g=600, Ds=[130, 77, 200], H=[l[0] for l in layers if l[0]][0], O=100, seed=21)
    a, b = _pair(prob, layers, batch_size=B, learning_rate=1e-3, seed=4242, loss=loss)
    for k in range(a.K):                                                   # Glorot init per layer: bit-exact
        for x, y in zip(a.get_weights(k), b.get_weights(k)):
            assert np.array_equal(x, y)
    for epoch in range(2):
        np.testing.assert_allclose(a.train_epoch(epoch), b.train_epoch(epoch), rtol=1e-4)
        np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=1e-4)
    assert a.step_count() == b.step_count()
    for k in range(a.K):
        for x, y in zip(a.get_weights(k), b.get_weights(k)):
            np.testing.assert_allclose(x, y, rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(a.predict(), b.predict(), rtol=1e-4, atol=1e-6)
    rows = prob["train"][:50]
    np.testing.assert_allclose(a.predict(rows), b.predict(rows), rtol=1e-4, atol=1e-6)
    a.close(); b.close()


def test_general_path_equals_tuned_kernels_on_default_architecture():
    """Same Philox streams, same init: the general path trains the default architecture like the tuned kernels (fp32 rounding apart)."""
    from deepimpute_amd.engine import HipEngine
    from helpers import load_problem
    prob = make_problem(n=400, g=500, Ds=[150, 90], H=256, O=128, seed=8)
    kw = dict(batch_size=64, learning_rate=1e-3, seed=31)
    a = load_problem(HipEngine, prob, dropout_rate=0.2, **kw)
    b, _ = _pair(prob, [(256, "relu", 0.2)], **kw)
    a.init_weights()
    for e in range(2):
        np.testing.assert_allclose(a.train_epoch(e), b.train_epoch(e), rtol=1e-4)
    np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=1e-4)
    np.testing.assert_allclose(a.predict(), b.predict(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("layers,B", [
    ([(256, "relu", 0.2)], 64),                      # the default shapes on the general path (bench.py --general)
    ([(512, "relu", 0.2)], 128),                     # deepImpute --hidden-neurons 512 --batch-size 128
    ([(300, "relu", 0.2), (128, "tanh", 0.1)], 100),
])
def test_general_path_at_configs2_widths_matches_general_oracle(layers, B):
    """The general path at the predictor widths of BASELINE configs[2] (D ~ 2 400, ragged; O = 512): the first layer's forward runs
    split-K over the long inner dimension (k-ranges summed by k_gen_splitk_fin, dropout from one Philox block per four elements) and
    both first-layer GEMMs read their batch rows in place from the X arena -- code the small-shape tests above never reach (their D is
    below one k-range).  Two epochs incl. a partial batch, against oracle/dimo_general.c."""
    prob = make_problem(n=2 * B + B // 3 + 60, g=3000, Ds=[2400, 2391, 700], H=layers[0][0], O=512, seed=23)
    a, b = _pair(prob, layers, batch_size=B, learning_rate=1e-3, seed=99, loss="wmse")
    for epoch in range(2):
        np.testing.assert_allclose(a.train_epoch(epoch), b.train_epoch(epoch), rtol=1e-4)
        np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=1e-4)
    assert a.step_count() == b.step_count()
    for k in range(a.K):
        for x, y in zip(a.get_weights(k), b.get_weights(k)):
            np.testing.assert_allclose(x, y, rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(a.predict(), b.predict(), rtol=1e-4, atol=1e-6)
    rows = prob["val"][:7]
    np.testing.assert_allclose(a.predict(rows), b.predict(rows), rtol=1e-4, atol=1e-6)
    a.close(); b.close()


@pytest.mark.parametrize("knob", ["gemm=0", "gfuse=0", "gs=2", "gs=5"])
def test_general_row_gemm_forms_agree(knob, monkeypatch):
    """Round 6: the batch-row GEMMs of aligned shapes run on k_gen_rowgemm (operands straight from global memory, the output layer's
    loss / dZ fused behind its forward, split-K by the grid's size).  Same problem with each choice forced the other way
    (DIMN_RES_TEST, read per launch): the LDS-staged k_gen_gemm for all of them, the loss in k_gen_output's own launch, other numbers
    of k-ranges -- same training to fp32 rounding of the sums' order."""
    prob = make_problem(n=330, g=3000, Ds=[2400, 1210, 700, 64], H=256, O=512, seed=5)
    kw = dict(batch_size=64, learning_rate=1e-3, seed=7, loss="wmse")
    out = []
    for env in (None, knob):
        if env: monkeypatch.setenv("DIMN_RES_TEST", env)
        else: monkeypatch.delenv("DIMN_RES_TEST", raising=False)
        a, = _pair(prob, [(256, "relu", 0.2)], classes=(_hip(),), **kw)
        losses = [a.train_epoch(ep) for ep in range(2)]
        out.append((np.asarray(losses), np.asarray(a.val_loss()), a.predict(), [w for k in range(a.K) for w in a.get_weights(k)]))
        a.close()
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=2e-6)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=2e-6)
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=2e-4, atol=2e-6)
    for x, y in zip(out[0][3], out[1][3]):
        np.testing.assert_allclose(x, y, rtol=1e-3, atol=2e-6)


def test_general_random_shapes_agree_between_gemm_forms():
    """tools/gen_shape_sweep.py: random widths (multiples of 4 and not), ragged predictor counts (the masked last chunk of K), batches 1 .. 200,
    one to three hidden layers, every loss -- the default kernels against DIMN_RES_TEST=gemm=0, one epoch + validation + prediction."""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_shape_sweep.py")
    env = {k: v for k, v in os.environ.items() if k != "DIMN_RES_TEST"}
    r = subprocess.run([sys.executable, tool, "16", "11"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all 16 cases agree" in r.stdout
