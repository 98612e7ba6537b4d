"""BASELINE configs[4] pieces on one GPU: the bfloat16 X arena (precision="bf16") and the host-streamed matrix
(set_matrix(streamed=True): row blocks through pinned buffers, the matrix never resident).  Parity definition of the
bf16 arena: fp32 arithmetic on predictor values rounded to bfloat16 (nearest even) when they are stored; targets,
weights, Adam state and accumulation are fp32 -- so the oracle in the same mode must agree at the fp32 tolerances."""
import os

import numpy as np
import pytest

from helpers import load_problem, make_problem

pytestmark = pytest.mark.gpu


def _hip():
    from deepimpute_amd.engine import HipEngine
    return HipEngine


def _oracle():
    from oracle.dimo import OracleEngine
    return OracleEngine


@pytest.mark.parametrize("H,O,B,Ds,p", [
    (256, 512, 64, [300, 150, 77], 0.2),    # ring B1F1 + fused second layer
    (300, 512, 64, [260], 0.35),            # CLI default hidden = 300 (shared-staging B1F1)
    (150, 100, 37, [97, 64, 33], 0.2),      # generic B1F1, ragged everything, partial batches
])
@pytest.mark.parametrize("resident", ["0", "1"])     # streaming kernels / register-resident epoch kernel (H = 256 only) on the bf16 arena
def test_bf16_x_arena_matches_oracle_on_rounded_inputs(H, O, B, Ds, p, resident, monkeypatch):
    if resident == "1" and H != 256:
        pytest.skip("the resident kernel takes H = 256")
    monkeypatch.setenv("DIMN_RESIDENT", resident)
    monkeypatch.setenv("DIMN_TRAIN_BF16", "0")        # (and no bf16 training GEMMs, should the fused second layer be chosen)
    monkeypatch.setenv("DIMN_PREDICT_BF16", "0")      # this test pins the arena alone: inference GEMMs in fp32 too
    prob = make_problem(n=330, g=700, Ds=Ds, H=H, O=O, seed=11)
    kw = dict(batch_size=B, dropout_rate=p, learning_rate=1e-3, seed=4242, precision="bf16")
    a = load_problem(_hip(), prob, **kw)
    b = load_problem(_oracle(), prob, **kw)
    c = load_problem(_oracle(), prob, **dict(kw, precision="fp32"))
    for e in (a, b, c):
        e.init_weights()
    a.set_profiling(True)
    for epoch in range(2):
        la, lb, lc = a.train_epoch(epoch), b.train_epoch(epoch), c.train_epoch(epoch)
        np.testing.assert_allclose(la, lb, rtol=1e-4)
        np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=1e-4)
    assert (a.get_timers()[7] == a.step_count()) == (resident == "1")
    assert not np.allclose(lb, lc, rtol=1e-5)         # the rounding of the inputs is visible (and small)
    np.testing.assert_allclose(lb, lc, rtol=2e-2)
    for k in range(a.K):
        for x, y, name in zip(a.get_weights(k), b.get_weights(k), ("W1", "b1", "W2", "b2")):
            np.testing.assert_allclose(x, y, rtol=1e-3, atol=2e-5, err_msg="%s k=%d" % (name, k))
    np.testing.assert_allclose(a.predict(), b.predict(), rtol=1e-4, atol=1e-6)
    for e in (a, b, c):
        e.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_streamed_matrix_equals_resident_matrix(precision):
    """The streamed hand-over (many row blocks: the block size is ~128 MB, so a wide matrix is used) gathers the same
    X_k / Y_k as the resident one: identical training and prediction, bit for bit."""
    rng = np.random.default_rng(3)
    n, g = 5000, 16000                                 # 320 MB matrix -> 3 streamed blocks
    norm = np.log1p(rng.poisson(1.5, size=(n, g))).astype(np.float32)
    Ds = [200, 120]
    pred = [rng.choice(g, D, replace=False).astype(np.int32) for D in Ds]
    targ = [rng.choice(g, 64, replace=False).astype(np.int32) for _ in Ds]
    train, val = np.arange(0, 1000, dtype=np.int32), np.arange(4000, 4200, dtype=np.int32)
    outs = []
    # resident; streamed with only the columns the sub-nets read packed into the bounce buffer (the default: 2 % of the genes here);
    # streamed as whole rows (DIMN_STREAM_PACK=0)
    # ... and, as rank 1 of 3 / rank 2 of 3 of a job whose ranks read ONE host copy, starting a third / two thirds into the row blocks
    # (dimn_set_stream_order: the blocks are independent, their order is free)
    for streamed, pack, order in ((False, None, None), (True, None, None), (True, "0", None), (True, None, (1, 3)), (True, "0", (2, 3))):
        if pack is None:
            os.environ.pop("DIMN_STREAM_PACK", None)
        else:
            os.environ["DIMN_STREAM_PACK"] = pack
        e = _hip()(Ds, 64, 64, batch_size=64, dropout_rate=0.2, learning_rate=1e-3, seed=5, precision=precision)
        for k in range(2):
            e.set_indices(k, pred[k], targ[k])
        if order is not None:
            e.set_stream_order(*order)
        e.set_matrix(norm, streamed=streamed)
        e.gather(True)
        e.set_split(train, val)
        e.init_weights()
        outs.append((e.train_epoch(0), e.val_loss(), e.predict(np.arange(0, n, 7, dtype=np.int32))))
        e.close()
    os.environ.pop("DIMN_STREAM_PACK", None)
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert np.array_equal(x, y)


def test_general_path_on_bf16_arena_matches_general_oracle_rounded():
    from deepimpute_amd.engine import HipGeneralEngine
    from oracle.dimo import GeneralOracleEngine
    prob = make_problem(n=400, g=500, Ds=[130, 77], H=96, O=100, seed=21)
    layers = [(96, "relu", 0.2), (48, "tanh", 0.0)]
    rounded = prob["norm"].view(np.uint32)
    rounded = (((rounded + 0x7FFF + ((rounded >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
    engines = []
    for cls, matrix, prec in ((HipGeneralEngine, prob["norm"], "bf16"),):
        e = cls(prob["Ds"], layers, prob["O"], batch_size=128, learning_rate=1e-3, seed=7, precision=prec)
        e.set_matrix(matrix)
        engines.append(e)
    # the general oracle has no precision switch: it gets pre-rounded predictors and unrounded targets through two matrices
    o = GeneralOracleEngine(prob["Ds"], layers, prob["O"], batch_size=128, learning_rate=1e-3, seed=7)
    g = prob["norm"].shape[1]
    both = np.hstack([rounded, prob["norm"]])          # columns [0, g): rounded (predictors); [g, 2g): exact (targets)
    o.set_matrix(both)
    a = engines[0]
    for k in range(2):
        a.set_indices(k, prob["pred"][k], prob["targ"][k])
        o.set_indices(k, prob["pred"][k], prob["targ"][k] + g)
    a.gather(True)
    for e in (a, o):
        e.set_split(prob["train"], prob["val"])
        e.init_weights()
    for epoch in range(2):
        np.testing.assert_allclose(a.train_epoch(epoch), o.train_epoch(epoch), rtol=1e-4)
    np.testing.assert_allclose(a.val_loss(), o.val_loss(), rtol=1e-4)
    np.testing.assert_allclose(a.predict(), o.predict(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("H,O,B,Ds", [(256, 512, 64, [300, 150, 77]), (300, 512, 64, [260]), (150, 100, 37, [97, 64, 33])])
def test_bf16_matrix_core_inference_matches_oracle_rounding(H, O, B, Ds, monkeypatch):
    """precision="bf16" in full: the arena in bf16 and model.predict / the validation pass on v_mfma_f32_16x16x16_bf16
    (k_predict_bf16: X, W1, the hidden activations and W2 as bfloat16 operands, fp32 accumulation).  The oracle restates
    exactly that rounding (infer_bf16); products of two bf16 values are exact in fp32, so what is left is the summation
    order -- and, rarely, a hidden activation that rounds to the neighbouring bf16 value (a 0.4 % step of one of the H
    terms of an output).  Stated tolerance: validation loss 5e-4 relative, imputed values 2e-3 relative + 2e-4 absolute;
    against the all-fp32 path the predictions move by < 2 % (the price of the format, not of the kernel)."""
    monkeypatch.setenv("DIMN_TRAIN_BF16", "0")        # this test pins INFERENCE on the bf16 matrix cores: training GEMMs stay fp32
    prob = make_problem(n=330, g=700, Ds=Ds, H=H, O=O, seed=11)
    kw = dict(batch_size=B, dropout_rate=0.2, learning_rate=1e-3, seed=4242)
    a = load_problem(_hip(), prob, precision="bf16", **kw)
    assert a.path_info()["train_bf16"] == 0
    b = load_problem(_oracle(), prob, precision="bf16", infer_bf16=True, **kw)
    c = load_problem(_hip(), prob, precision="fp32", **kw)
    for e in (a, b, c):
        e.init_weights()
    for epoch in range(2):
        np.testing.assert_allclose(a.train_epoch(epoch), b.train_epoch(epoch), rtol=1e-4)      # training GEMMs: fp32 matrix cores
        np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=5e-4)
        c.train_epoch(epoch)
    pa, pb, pc = a.predict(), b.predict(), c.predict()
    np.testing.assert_allclose(pa, pb, rtol=2e-3, atol=2e-4)
    rows = prob["val"]
    np.testing.assert_allclose(a.predict(rows), pa[rows], rtol=0, atol=0)
    assert np.abs(pa - pc).max() / np.abs(pc).max() < 2e-2
    for e in (a, b, c):
        e.close()


@pytest.mark.parametrize("H,O,B,Ds,n,act", [
    (256, 101, 64, [300, 77], 330, "tanh"),       # the instantiation WITH the activation switch and the scalar output stores (O % 4 != 0)
    (256, 512, 64, [10, 40], 200, "relu"),        # one-chunk and three-chunk sub-nets: steps, pairs and k-halves that do not exist
    (64, 32, 16, [129], 100, "sigmoid"),          # fewer rows than one 128-row workgroup, hidden width below a wave's 64 columns, an odd step count
    (256, 512, 64, [520], 300, "relu"),           # nine 64-deep steps: an odd count above one pair
])
def test_bf16_matrix_core_inference_edge_shapes(H, O, B, Ds, n, act, monkeypatch):
    """k_predict_bf16's round-4 loop (64-deep steps in pairs through the LDS DMA, chunks past a sub-net's last one fetched as zeros,
    hand-counted waits) and both of its instantiations, at the shapes where a count could be off; tolerances as above."""
    monkeypatch.setenv("DIMN_TRAIN_BF16", "0")
    prob = make_problem(n=n, g=700, Ds=Ds, H=H, O=O, seed=5)
    kw = dict(batch_size=B, dropout_rate=0.2, learning_rate=1e-3, seed=77, activation=act)
    a = load_problem(_hip(), prob, precision="bf16", **kw)
    b = load_problem(_oracle(), prob, precision="bf16", infer_bf16=True, **kw)
    for e in (a, b):
        e.init_weights()
    np.testing.assert_allclose(a.train_epoch(0), b.train_epoch(0), rtol=1e-4)
    np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=5e-4)
    pa, pb = a.predict(), b.predict()
    assert np.isfinite(pa).all()
    np.testing.assert_allclose(pa, pb, rtol=2e-3, atol=2e-4)
    rows = prob["val"]
    np.testing.assert_allclose(a.predict(rows), pa[rows], rtol=0, atol=0)
    for e in (a, b):
        e.close()


def test_bf16_matrix_core_training_of_the_second_layer(monkeypatch):
    """precision="bf16" on the fused second-layer kernel (the tile pipeline k_mid_pipe<BF>): Z = Dd W2, gW2 = Dd^T dZ and dD = dZ W2^T take
    bf16 operands (rounded to nearest even in registers, fp32 accumulation, fp32 master weights and Adam state).  The oracle
    restates the rounding (train_bf16); what is left is the summation order and, rarely, an operand that rounds to the
    neighbouring bf16 value.  Stated tolerance: training / validation loss 1e-3 relative, imputed values 5e-3 relative +
    5e-4 absolute after two epochs; DIMN_TRAIN_BF16=0 keeps the fp32 matrix cores."""
    monkeypatch.setenv("DIMN_MID", "1")
    monkeypatch.setenv("DIMN_RESIDENT", "0")
    prob = make_problem(n=330, g=700, Ds=[300, 150, 77], H=256, O=512, seed=11)
    kw = dict(batch_size=64, dropout_rate=0.2, learning_rate=1e-3, seed=4242)
    a = load_problem(_hip(), prob, precision="bf16", **kw)
    assert a.training_precision == "bf16"
    assert a.path_info()["mid_keep"] == 2, a.path_info()
    b = load_problem(_oracle(), prob, precision="bf16", infer_bf16=True, train_bf16=True, **kw)
    c = load_problem(_oracle(), prob, precision="bf16", infer_bf16=True, **kw)           # fp32 training GEMMs
    for e in (a, b, c):
        e.init_weights()
    for epoch in range(2):
        la, lb, lc = a.train_epoch(epoch), b.train_epoch(epoch), c.train_epoch(epoch)
        np.testing.assert_allclose(la, lb, rtol=1e-3)
        np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=1e-3)
    np.testing.assert_allclose(a.predict(), b.predict(), rtol=5e-3, atol=5e-4)
    # the rounding is really there: the HIP path sits closer to the oracle that rounds than to the one that does not
    w_a, w_b, w_c = a.get_weights(0)[2], b.get_weights(0)[2], c.get_weights(0)[2]
    assert np.abs(w_a - w_b).mean() < 0.5 * np.abs(w_a - w_c).mean()
    monkeypatch.setenv("DIMN_TRAIN_BF16", "0")
    d = load_problem(_hip(), prob, precision="bf16", **kw)
    assert d.training_precision == "fp32"
    for e in (a, b, c, d):
        e.close()


@pytest.mark.parametrize("groups", ["1", "2"])
def test_bf16_matrix_core_training_on_the_resident_kernel(groups, monkeypatch):
    """precision="bf16" on the register-resident epoch kernel (k_epoch_resident<.., BF>): EVERY training GEMM of the step -- the
    first layer's forward and W1 gradient, the second layer's three -- takes bf16 operands (the four k-slot values of four
    fp32 matrix instructions rounded to nearest even in registers = one v_mfma_f32_16x16x16_bf16), fp32 accumulation, fp32
    master weights and Adam state; also in two groups of sub-nets (one epoch launch each).  The oracle restates the rounding
    (train_bf16 = 2); tolerances as for the fused second layer: losses 1e-3, imputed values 5e-3 + 5e-4 after two epochs."""
    monkeypatch.setenv("DIMN_RESIDENT", "1")
    monkeypatch.setenv("DIMN_RES_TEST", "groups=" + groups)
    prob = make_problem(n=330, g=700, Ds=[300, 150, 77, 210], H=256, O=512, seed=11)
    kw = dict(batch_size=64, dropout_rate=0.2, learning_rate=1e-3, seed=4242)
    a = load_problem(_hip(), prob, precision="bf16", **kw)
    info = a.path_info()
    assert info["path"] == "resident" and info["train_bf16"] == 2 and info["resident_groups"] == int(groups) and a.training_precision == "bf16"
    b = load_problem(_oracle(), prob, precision="bf16", infer_bf16=True, train_bf16=2, **kw)
    c = load_problem(_oracle(), prob, precision="bf16", infer_bf16=True, **kw)           # fp32 training GEMMs
    for e in (a, b, c):
        e.init_weights()
    a.set_profiling(True)
    for epoch in range(2):
        la, lb, lc = a.train_epoch(epoch), b.train_epoch(epoch), c.train_epoch(epoch)
        np.testing.assert_allclose(la, lb, rtol=1e-3)
        np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=1e-3)
    assert a.get_timers()[7] == a.step_count()
    np.testing.assert_allclose(a.predict(), b.predict(), rtol=5e-3, atol=5e-4)
    # the rounding is really there: the HIP path sits closer to the oracle that rounds than to the one that does not
    w_a, w_b, w_c = a.get_weights(0)[0], b.get_weights(0)[0], c.get_weights(0)[0]
    assert np.abs(w_a - w_b).mean() < 0.5 * np.abs(w_a - w_c).mean()
    for e in (a, b, c):
        e.close()


def test_multinet_streamed_and_bf16_through_the_shell(tmp_path):
    """MultiNet(stream_matrix=True) imputes exactly what the resident hand-over does; MultiNet(precision="bf16") stays close
    to fp32 (same early-stopping epoch on this problem, held-out correlation within 1e-2)."""
    import pandas as pd
    from deepimpute_amd.multinet import MultiNet
    rng = np.random.default_rng(0)
    n, g = 300, 700
    u, v = rng.normal(size=(n, 6)), rng.normal(size=(g, 6))
    counts = rng.poisson(np.exp(0.6 * (u @ v.T) / np.sqrt(6) + rng.normal(0.3, 0.8, size=g))).astype(np.float64)
    counts[:, :5] += rng.poisson(20, size=(n, 5))
    raw = pd.DataFrame(counts, index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
    kw = dict(sub_outputdim=128, seed=123, ncores=2, verbose=0, max_epochs=12, learning_rate=1e-3)
    res = MultiNet(output_prefix=str(tmp_path / "a"), **kw).fit(raw)
    stm = MultiNet(output_prefix=str(tmp_path / "b"), stream_matrix=True, **kw).fit(raw)
    assert res.history == stm.history
    assert np.array_equal(res.predict(raw).values, stm.predict(raw).values)
    b16 = MultiNet(output_prefix=str(tmp_path / "c"), precision="bf16", stream_matrix=True, **kw).fit(raw)
    assert b16._engine.precision == "bf16" and b16.trained_epochs == res.trained_epochs
    assert abs(float(b16.test_metrics["correlation"]) - float(res.test_metrics["correlation"])) < 1e-2
    out = b16.predict(raw)
    assert np.isfinite(out.values).all() and np.array_equal(out.values[counts > 0], counts[counts > 0])
    fresh = MultiNet(output_prefix=str(tmp_path / "c"), precision="bf16", **kw)            # reload on a bf16 engine
    fresh.predictors, fresh.targets = b16.predictors, b16.targets
    np.testing.assert_allclose(fresh.predict(raw).values, out.values, rtol=1e-6)
