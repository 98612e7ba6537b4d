"""GPU parity against the oracle at the real shapes of BASELINE.json configs[1] (5k x 5k, K = 10, D_k ~ 1935, H = 256, O = 512),
configs[2] (K = 40, D_k ~ 2400), one rank's share of configs[3] (5 sub-nets, resident kernel) and of configs[4] (g = 30 000,
8 sub-nets, bf16, streamed), and a
long-horizon drift check of the fast-math pieces of the HIP path (1-ulp v_rcp/v_sqrt in Adam, hardware exp/log in
the training softplus) against the fp64 build of the oracle.  Run on the MI355X box: pytest -m gpu."""
import os

import numpy as np
import pytest

from helpers import find_relu_flip_candidates, oracle_with_inverted_gates, relu_flip_units

pytestmark = pytest.mark.gpu


def _hip():
    from deepimpute_amd.engine import HipEngine
    return HipEngine


def _oracle():
    from oracle.dimo import OracleEngine
    return OracleEngine


def _cfg2_problem():
    import bench
    cfg = bench.CONFIGS["cfg2"]
    n, g = cfg["n"], cfg["g"]
    norm = bench.synth_counts(n, g, seed=0)
    targets, preds = bench.synth_indices(g, cfg["O"], seed=0)
    train, val = bench.split_rows(n, seed=0)
    return cfg, norm, targets, preds, train, val


@pytest.mark.parametrize("mid", ["1", "0", "R"])      # fused second layer / two-kernel second layer (DIMN_MID) / what the library picks: two resident groups of 5
def test_cfg2_shapes_match_oracle(mid, monkeypatch):
    """configs[1] at its real shapes: 4 optimiser steps (the last one partial), the validation pass over the 5 %
    split and predict of 256 cells, HIP vs oracle, tolerances of test_two_epochs_match_oracle."""
    if mid == "R":
        monkeypatch.setenv("DIMN_RESIDENT", "1")
    else:
        monkeypatch.setenv("DIMN_RESIDENT", "0")
        monkeypatch.setenv("DIMN_MID", mid)
    cfg, norm, targets, preds, train, val = _cfg2_problem()
    K = targets.shape[0]
    assert K == 10 and 1800 < min(map(len, preds)) and max(map(len, preds)) < 2100
    engines = []
    for cls in (_hip(), _oracle()):
        e = cls([len(p) for p in preds], cfg["H"], cfg["O"], batch_size=cfg["B"], dropout_rate=0.2, learning_rate=1e-3, seed=1234)
        e.set_matrix(norm)
        for k in range(K):
            e.set_indices(k, preds[k], targets[k])
        e.gather(True)
        e.set_split(train[:3 * 64 + 21], val)
        e.init_weights()
        engines.append(e)
    a, b = engines
    a.set_profiling(True)
    rows = np.arange(7, 7 + 256 * 19, 19, dtype=np.int32)
    kw = dict(batch_size=cfg["B"], dropout_rate=0.2, learning_rate=1e-3, seed=1234)
    # (R: on this problem the resident kernel's summation order puts one pre-activation of sub-net 3 -- hidden unit 52, third
    #  batch, a = -6.5e-7 in the fp64 replay, a tenth of the fp32 reordering bound -- on the other side of zero: see
    #  _compare_with_oracle for how the test treats that)
    _compare_with_oracle(a, b, norm, preds, targets, list(range(K)), train[:3 * 64 + 21], val, 4, cfg, kw, rows)
    if mid == "R":
        assert a.get_timers()[7] == 4                          # the resident kernel ran (two launches of five sub-nets per epoch)
    a.close(); b.close()


def _compare_with_oracle(a, b, norm, preds, targets, ks, train, val, steps, cfg, kw, rows, tol=None, oracle_kw=None):
    """One epoch of `steps` optimiser steps (the last one partial) on both engines, then validation, weights and predict at the
    tolerances of test_two_epochs_match_oracle, sub-net by sub-net.

    Two fp32 paths that sum a first-layer dot product in different orders can put a pre-activation that is zero to fp32
    precision on different sides of zero; relu'(a) is discontinuous there, so that one (row, unit) takes a different Adam
    step and everything downstream in that sub-net moves with it.  The test does not grant a budget for that: a sub-net
    with a first-layer column outside the weight tolerance must (1) show, in the fp64 replay, a pre-activation of a
    flagged unit within fp32 reordering error of zero (helpers.find_relu_flip_candidates), and (2) agree with the oracle
    at the FULL tolerances once the oracle takes exactly that gate on the other side (dimo_invert_gate)."""
    import itertools
    tol = dict(dict(loss=1e-4, w_rtol=1e-3, w_atol=2e-5, p_rtol=1e-4, p_atol=1e-6, weights=True), **(tol or {}))
    O = cfg["O"]
    la, lb = a.train_epoch(0), b.train_epoch(0)
    assert a.step_count() == b.step_count() == steps
    va, vb = a.val_loss(), b.val_loss()
    pa, pb = a.predict(rows), b.predict(rows)

    def same(i, o, j, lo, vo, po):
        np.testing.assert_allclose(la[i], lo, rtol=tol["loss"], err_msg="train loss k=%d" % ks[i])
        np.testing.assert_allclose(va[i], vo, rtol=tol["loss"], err_msg="val loss k=%d" % ks[i])
        for x, y, name in zip(a.get_weights(i), o.get_weights(j), ("W1", "b1", "W2", "b2")):
            if tol["weights"]:
                np.testing.assert_allclose(x, y, rtol=tol["w_rtol"], atol=tol["w_atol"], err_msg="%s k=%d" % (name, ks[i]))
        np.testing.assert_allclose(pa[:, i * O:(i + 1) * O], po, rtol=tol["p_rtol"], atol=tol["p_atol"], err_msg="predict k=%d" % ks[i])

    flipped = {}
    for i in range(len(ks)):
        units = relu_flip_units(a, b, i, tol["w_rtol"], tol["w_atol"]) if tol["weights"] else np.zeros(0, int)
        if units.size == 0:
            same(i, b, i, lb[i], vb[i], pb[:, i * O:(i + 1) * O])
            continue
        assert len(flipped) < 2, "more than two sub-nets off: not the rare event this path is for"
        args = (_oracle(), norm, preds[ks[i]], targets[ks[i]], ks[i])
        cands = find_relu_flip_candidates(*args, units, train, steps, cfg["H"], O, **kw, **(oracle_kw or {}))
        if not cands:
            # No gate at fp32 noise level.  The other way two fp32 evaluations drift apart in a few steps is Adam's normalisation of a small
            # gradient (step = lr m / (sqrt(v) + eps): the rounding of g is amplified by 1 / |g|) -- then it is the plain-loop fp32 ORACLE
            # that may be the one further from the truth.  Accepted only on evidence: the float64 oracle of this sub-net on the same streams,
            # and the HIP path at most as far from it as the fp32 oracle is (x 2), array by array (first seen: hidden 272, sub-net 37, unit
            # 249 -- HIP 7e-6 from the float64 weights, the fp32 oracle 2.6e-5, one weight 3.3e-5 apart against 3.2e-5 allowed).
            o64, l64 = oracle_with_inverted_gates(*args, train, val, cfg["H"], O, [], fp64=True, **kw, **(oracle_kw or {}))
            try:
                p64 = o64.predict(rows)
                pairs = list(zip(a.get_weights(i), b.get_weights(i), o64.get_weights(0), ("W1", "b1", "W2", "b2")))
                pairs.append((pa[:, i * O:(i + 1) * O], pb[:, i * O:(i + 1) * O], p64, "predict"))
                pairs.append((np.asarray(la[i]), np.asarray(lb[i]), np.asarray(l64[0]), "train loss"))
                for x, y, z, name in pairs:
                    eh, eo = float(np.max(np.abs(x - z))), float(np.max(np.abs(y - z)))
                    assert eh <= 2.0 * eo + 1e-7, "sub-net %d, %s: units %s miss the tolerance, no relu flip, and the HIP path is %.3e from the " \
                        "float64 oracle where the fp32 oracle is %.3e" % (ks[i], name, units.tolist(), eh, eo)
                print("sub-net %d: units %s outside the weight tolerance against the fp32 oracle, which is the one further from the float64 "
                      "oracle (no relu gate involved)" % (ks[i], units.tolist()))
                flipped[ks[i]] = ("fp32 oracle further from float64 than the HIP path",)
            finally:
                o64.close()
            continue
        last = None
        for inv in [c for c in itertools.chain(((x,) for x in cands[:4]), itertools.combinations(cands[:4], 2))]:
            o, lo = oracle_with_inverted_gates(*args, train, val, cfg["H"], O, [r[:4] for r in inv], **kw, **(oracle_kw or {}))
            try:
                same(i, o, 0, lo[0], o.val_loss()[0], o.predict(rows))
                flipped[ks[i]] = inv
                break
            except AssertionError as e:
                last = e
            finally:
                o.close()
        else:
            raise last
    if flipped:
        print("relu flips (mechanism asserted, oracle re-run with the gate on the other side): (epoch, step, batch position, unit, a, bound)", flipped)
    return flipped


def _cfg3_sample(n_cells):
    """The predictor / target lists of the 50k x 20k job (bench.synth_indices(20000, 512): K = 40, D_k ~ 2 390-2 420) over the
    first n_cells cells of a matrix from the same generator (the oracle's cost is per row; the shapes that select kernels --
    K, D_k, H, O, batch -- are the full job's)."""
    import bench
    cfg = dict(bench.CONFIGS["cfg3"])
    norm = bench.synth_counts(n_cells, cfg["g"], seed=0)
    targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=0)
    return cfg, norm, targets, preds


def _load(cls, cfg, norm, preds, targets, ks, train, val, streamed=False, **kw):
    e = cls([len(preds[k]) for k in ks], cfg["H"], cfg["O"], subnet_offset=ks[0], **kw)
    for i, k in enumerate(ks):
        e.set_indices(i, preds[k], targets[k])
    e.set_matrix(norm, **({"streamed": True} if streamed else {}))
    e.gather(True)
    e.set_split(train, val)
    e.init_weights()
    return e


@pytest.mark.parametrize("mid", [None, "0"])      # what the library picks (the fused tile pipeline) / the two-kernel second layer forced (k_mid_fwd<16, 6> at 40 sub-nets + k_mid_bwd)
def test_cfg3_shapes_match_oracle(mid, monkeypatch):
    """BASELINE configs[2] -- the config the metric is quoted on -- at its real shapes against the ORACLE: all K = 40 sub-nets
    with the D_k of the 50k x 20k job, H = 256, O = 512, batch 64, on whatever path the library picks by itself (no DIMN_*
    switch: ring B1F1 + the fused second layer in 6-tile slices): 3 full + 1 partial optimiser step, validation over 250 cells,
    predict of 256 cells.  (Four oracle steps of this size cost ~3 s on the box's cores.)"""
    if mid is not None:
        monkeypatch.setenv("DIMN_MID", mid)
    cfg, norm, targets, preds = _cfg3_sample(2048)
    K = targets.shape[0]
    assert K == 40 and 2300 < min(map(len, preds)) and max(map(len, preds)) < 2500
    train = np.arange(0, 3 * 64 + 21, dtype=np.int32) * 7 % 1700
    val = np.arange(1700, 1950, dtype=np.int32)
    rows = np.arange(3, 3 + 256 * 8, 8, dtype=np.int32) % 2048
    kw = dict(batch_size=cfg["B"], dropout_rate=0.2, learning_rate=1e-3, seed=1234)
    a = _load(_hip(), cfg, norm, preds, targets, list(range(K)), train, val, **kw)
    b = _load(_oracle(), cfg, norm, preds, targets, list(range(K)), train, val, **kw)
    a.set_profiling(True)
    assert a.path_info()["mid_fused"] == (1 if mid is None else 0), a.path_info()
    _compare_with_oracle(a, b, norm, preds, targets, list(range(K)), train, val, 4, cfg, kw, rows)
    t = a.get_timers()
    assert t[7] == 0 and t[1] >= 1            # the streaming kernels ran (step_launch times one step in eight), no resident launch
    a.close(); b.close()


def test_cfg3_shapes_hidden_300_match_oracle():
    """The CLI's default hidden width (parser.py:62-66; 300 = 19 hidden tiles, zero-padded to 20) at the shapes of the 50k x 20k
    job, all K = 40 sub-nets, against the ORACLE: 3 full + 1 partial optimiser step, validation, predict -- on what the library
    picks: the hidden tiles of a D-slice over two workgroups of 10 waves x 1 tile with a four-set register ring
    (k_w1_update_fwd_ring<10, 1, 4>, grid.y = 2, 128 D-slices), the second layer's forward in 6 slices of six output tiles (12 waves,
    one round of workgroups).  (The 10 x 2 two-set kernel and the 8-slice forward of rounds 2-4 were retired in round 6.)"""
    cfg, norm, targets, preds = _cfg3_sample(2048)
    cfg = dict(cfg, H=300)
    K = targets.shape[0]
    train = np.arange(0, 3 * 64 + 21, dtype=np.int32) * 7 % 1700
    val = np.arange(1700, 1950, dtype=np.int32)
    rows = np.arange(3, 3 + 256 * 8, 8, dtype=np.int32) % 2048
    kw = dict(batch_size=cfg["B"], dropout_rate=0.2, learning_rate=1e-3, seed=1234)
    a = _load(_hip(), cfg, norm, preds, targets, list(range(K)), train, val, **kw)
    b = _load(_oracle(), cfg, norm, preds, targets, list(range(K)), train, val, **kw)
    info = a.path_info()
    assert info["path"] == "streaming" and info["mid_fused"] == 0 and info["first_layer"] == 3, info
    _compare_with_oracle(a, b, norm, preds, targets, list(range(K)), train, val, 4, cfg, kw, rows)
    a.close(); b.close()


@pytest.mark.parametrize("H", [128, 192, 208, 272, 384])
def test_cfg3_shapes_other_hidden_widths_match_oracle(H):
    """Hidden widths other than 256 / 300 at the shapes of the 50k x 20k job (K = 40, D ~ 2 400) against the ORACLE: every width of 8 .. 24
    hidden tiles takes the first-layer ring kernel with one tile per wave and four register sets (round 5) -- 8 tiles as two workgroups
    of 8 waves per CU, 12 / 13 tiles as 12 / 13 waves, 17 tiles (272) padded to 18 = two halves of 9 waves, 24 tiles (384) as two halves
    of 12 -- and the two-kernel second layer.  3 full + 1 partial optimiser step, validation, predict."""
    cfg, norm, targets, preds = _cfg3_sample(2048)
    cfg = dict(cfg, H=H)
    K = targets.shape[0]
    train = np.arange(0, 3 * 64 + 21, dtype=np.int32) * 7 % 1700
    val = np.arange(1700, 1950, dtype=np.int32)
    rows = np.arange(3, 3 + 256 * 8, 8, dtype=np.int32) % 2048
    kw = dict(batch_size=cfg["B"], dropout_rate=0.2, learning_rate=1e-3, seed=1234)
    a = _load(_hip(), cfg, norm, preds, targets, list(range(K)), train, val, **kw)
    b = _load(_oracle(), cfg, norm, preds, targets, list(range(K)), train, val, **kw)
    info = a.path_info()
    assert info["path"] == "streaming" and info["mid_fused"] == 0 and info["first_layer"] == 3, info
    _compare_with_oracle(a, b, norm, preds, targets, list(range(K)), train, val, 4, cfg, kw, rows)
    a.close(); b.close()


def test_cfg3_shapes_hidden_300_bf16_arena_match_oracle():
    """The same shapes with precision bf16: the four-set ring over two hidden halves reading a bf16 X arena (its other
    instantiation), fp32 training GEMMs on this path (no fused second layer at 20 hidden tiles), inference on k_predict_bf16's
    sibling for hidden widths beyond 256 -- against the oracle in the matching rounding modes, tolerances of DESIGN section 3b."""
    cfg, norm, targets, preds = _cfg3_sample(2048)
    cfg = dict(cfg, H=300)
    K = targets.shape[0]
    train = np.arange(0, 3 * 64 + 21, dtype=np.int32) * 7 % 1700
    val = np.arange(1700, 1950, dtype=np.int32)
    rows = np.arange(3, 3 + 256 * 8, 8, dtype=np.int32) % 2048
    kw = dict(batch_size=cfg["B"], dropout_rate=0.2, learning_rate=1e-3, seed=1234, precision="bf16")
    a = _load(_hip(), cfg, norm, preds, targets, list(range(K)), train, val, **kw)
    info = a.path_info()
    assert info["path"] == "streaming" and info["first_layer"] == 3 and info["train_bf16"] == 0, info
    b = _load(_oracle(), cfg, norm, preds, targets, list(range(K)), train, val, infer_bf16=True, train_bf16=0, **kw)
    _compare_with_oracle(a, b, norm, preds, targets, list(range(K)), train, val, 4, cfg, kw, rows,
                         tol=dict(loss=5e-4, p_rtol=2e-3, p_atol=2e-4), oracle_kw=dict(infer_bf16=True, train_bf16=0))
    a.close(); b.close()


@pytest.mark.parametrize("split,erows", [("0", "0"), ("1", "0"), ("0", "1")])
# split: tile order of the kernel's loop, alternating / all gradient tiles first; erows: the epoch's rows copied into visiting order before the
# launch (what the library does for large arenas) -- both through DIMN_RES_TEST="split=..,erows=.."
def test_cfg4_8gpu_share_resident_matches_oracle(split, erows, monkeypatch):
    """BASELINE configs[3], one rank's share of the 8-GPU job (sub-nets 10-14 of the 40, global Philox keys) on the path the
    library picks for it -- the register-resident epoch kernel -- against the ORACLE (not against the streaming kernels): 5 full
    + 1 partial optimiser step, validation, predict; two epochs' worth of hand-off slots (t % 3, t % 2) are cycled."""
    monkeypatch.setenv("DIMN_RES_TEST", "split=%s,erows=%s" % (split, erows))
    cfg, norm, targets, preds = _cfg3_sample(2048)
    ks = list(range(10, 15))
    train = (np.arange(0, 5 * 64 + 33, dtype=np.int32) * 5) % 1800
    val = np.arange(1800, 2048, dtype=np.int32)
    rows = np.arange(0, 2048, 8, dtype=np.int32)
    kw = dict(batch_size=cfg["B"], dropout_rate=0.2, learning_rate=1e-3, seed=1234)
    a = _load(_hip(), cfg, norm, preds, targets, ks, train, val, **kw)
    b = _load(_oracle(), cfg, norm, preds, targets, ks, train, val, **kw)
    a.set_profiling(True)
    _compare_with_oracle(a, b, norm, preds, targets, ks, train, val, 6, cfg, kw, rows)
    assert a.get_timers()[7] == 6             # every step inside the resident launch
    a.close(); b.close()


@pytest.mark.parametrize("split,erows", [("0", "0"), ("1", "0"), ("0", "1")])         # (at the full 1M cells the library picks 0 / 1: 55 GB of rows)
def test_cfg5_share_shapes_match_oracle(split, erows, monkeypatch):
    """BASELINE configs[4], one rank's share at its real shapes: g = 30 000 genes, 8 of the K = 59 sub-nets (D_k ~ 2 450 from
    bench.synth_indices(30000, 512)), precision bf16 (bf16 X arena, bf16 matrix cores), the matrix STREAMED from host memory
    (three ~128 MB row blocks), on the path the library picks; 3 full + 1 partial step, validation, predict against the oracle
    in the matching rounding modes at the tolerances DESIGN section 3b states for them."""
    import bench
    monkeypatch.setenv("DIMN_RES_TEST", "split=%s,erows=%s" % (split, erows))
    cfg = dict(bench.CONFIGS["cfg5"])
    n = 3072
    norm = bench.synth_counts(n, cfg["g"], seed=0)                       # 369 MB -> 3 streamed blocks
    targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=0)
    assert targets.shape[0] == 59
    ks = list(range(8, 16))                                              # rank 1 of 8: 59 = 8+8+8+7+7+7+7+7
    assert 2300 < min(len(preds[k]) for k in ks) and max(len(preds[k]) for k in ks) < 2600
    train = (np.arange(0, 3 * 64 + 21, dtype=np.int32) * 11) % 2700
    val = np.arange(2700, 2950, dtype=np.int32)
    rows = np.arange(0, n, 12, dtype=np.int32)
    kw = dict(batch_size=cfg["B"], dropout_rate=0.2, learning_rate=1e-4, seed=1234, precision="bf16")      # the config's own learning rate
    a = _load(_hip(), cfg, norm, preds, targets, ks, train, val, streamed=True, **kw)
    bf_train = a.path_info()["train_bf16"]            # 0: fp32 training GEMMs, 1: the second layer's on bf16 operands (fused kernel), 2: all (resident kernel)
    b = _load(_oracle(), cfg, norm, preds, targets, ks, train, val, infer_bf16=True, train_bf16=bf_train, **kw)
    # bf16 inference operands 5e-4 on losses / 2e-3 + 2e-4 on imputed values; bf16 training operands 1e-3 / 5e-3 + 5e-4
    # (bf16 training operands: single weights are not compared -- where a gradient is at the rounding noise of its bf16 operands
    #  Adam's first steps move the weight by +-lr either way; the stated tolerances are on losses and imputed values)
    tol = dict(loss=1e-3, p_rtol=5e-3, p_atol=5e-4, weights=False) if bf_train else dict(loss=5e-4, p_rtol=2e-3, p_atol=2e-4)
    _compare_with_oracle(a, b, norm, preds, targets, ks, train, val, 4, cfg, kw, rows, tol=tol, oracle_kw=dict(infer_bf16=True, train_bf16=bf_train))
    print("cfg5 share ran on", a.path_info())
    a.close(); b.close()


@pytest.mark.parametrize("resident", ["0", "1"])     # streaming kernels / register-resident epoch kernel
def test_long_horizon_drift_vs_fp64(resident, monkeypatch):
    """>= 500 optimiser steps (K = 2, D = 300, H = 256, O = 512): the HIP path and the fp32 oracle are each compared
    with the fp64 oracle on the same Philox streams; the HIP path's error must not exceed the plain-loop fp32
    oracle's by more than 2x (plus an absolute floor of a few fp32 ulps of the quantities compared)."""
    from helpers import make_problem, load_problem
    monkeypatch.setenv("DIMN_RESIDENT", resident)
    prob = make_problem(n=1100, g=900, Ds=[300, 300], H=256, O=512, seed=23, val_frac=0.05)
    kw = dict(batch_size=64, dropout_rate=0.2, learning_rate=1e-4, seed=77)
    hip = load_problem(_hip(), prob, **kw)
    o32 = load_problem(_oracle(), prob, **kw)
    o64 = load_problem(_oracle(), prob, fp64=True, **kw)
    for e in (hip, o32, o64):
        e.init_weights()
    steps = 0
    epoch = 0
    while steps < 500:
        lh, l32, l64 = hip.train_epoch(epoch), o32.train_epoch(epoch), o64.train_epoch(epoch)
        steps = o64.step_count()
        epoch += 1
        err_h, err_o = np.abs(lh - l64) / l64, np.abs(l32 - l64) / l64
        assert (err_h <= 2 * err_o + 2e-6).all(), (epoch, err_h, err_o)
    assert hip.step_count() == steps >= 500
    vh, v32, v64 = hip.val_loss(), o32.val_loss(), o64.val_loss()
    assert (np.abs(vh - v64) / v64 <= 2 * np.abs(v32 - v64) / v64 + 2e-6).all()
    ph, p32, p64 = hip.predict(), o32.predict(), o64.predict()
    eh, eo = np.abs(ph - p64), np.abs(p32 - p64)
    assert eh.max() <= 2 * eo.max() + 1e-6, (eh.max(), eo.max())
    assert np.sqrt((eh ** 2).mean()) <= 2 * np.sqrt((eo ** 2).mean()) + 1e-7
    for k in range(2):
        for x, y, z, name in zip(hip.get_weights(k), o32.get_weights(k), o64.get_weights(k), ("W1", "b1", "W2", "b2")):
            dh, do = np.abs(x - z), np.abs(y - z)
            assert np.sqrt((dh ** 2).mean()) <= 2 * np.sqrt((do ** 2).mean()) + 1e-8, (name, k)
    for e in (hip, o32, o64):
        e.close()


def test_resident_epoch_kernel_matches_streaming_at_8gpu_share():
    """BASELINE configs[3], one rank's share (5 of the 40 sub-nets of the 50k x 20k job): the register-resident epoch
    kernel against the streaming kernels on the same Philox streams -- 41 optimiser steps incl. a partial batch, then
    validation and predict.  Also: two resident runs are bit-identical (a stale hand-off between workgroups would
    show up as run-to-run differences)."""
    import os
    import bench
    cfg = bench.CONFIGS["cfg3"]
    n, g = cfg["n"], cfg["g"]
    norm = bench.synth_counts(n, g, seed=0)
    targets, preds = bench.synth_indices(g, cfg["O"], seed=0)
    rng = np.random.default_rng(1)
    train = np.sort(rng.choice(n, 64 * 40 + 17, replace=False)).astype(np.int32)
    val = np.setdiff1d(np.arange(n, dtype=np.int32), train)[:1000].astype(np.int32)
    rows = np.arange(0, n, 97, dtype=np.int32)

    def run(resident, k0=10, k1=15):
        old = os.environ.get("DIMN_RESIDENT")
        os.environ["DIMN_RESIDENT"] = resident
        try:
            e = _hip()([len(preds[k]) for k in range(k0, k1)], cfg["H"], cfg["O"], batch_size=64, dropout_rate=0.2,
                       learning_rate=1e-4, seed=1234, subnet_offset=k0)
        finally:
            if old is None:
                del os.environ["DIMN_RESIDENT"]
            else:
                os.environ["DIMN_RESIDENT"] = old
        e.set_matrix(norm)
        for i, k in enumerate(range(k0, k1)):
            e.set_indices(i, preds[k], targets[k])
        e.gather(True)
        e.set_split(train, val)
        e.init_weights()
        out = [e.train_epoch(0), e.train_epoch(1), e.val_loss(), e.predict(rows), e.get_weights(2), e.get_adam_state(2, 1)]
        assert e.step_count() == 82
        e.close()
        return out

    a, b, c = run("1"), run("1"), run("0")
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)
    for x, y in zip(a[4] + a[5], b[4] + b[5]):
        assert np.array_equal(x, y)
    np.testing.assert_allclose(a[0], c[0], rtol=1e-5)
    np.testing.assert_allclose(a[1], c[1], rtol=1e-5)
    np.testing.assert_allclose(a[2], c[2], rtol=1e-5)
    np.testing.assert_allclose(a[3], c[3], rtol=1e-4, atol=1e-6)
    for x, y, name in zip(a[4], c[4], ("W1", "b1", "W2", "b2")):
        np.testing.assert_allclose(x, y, rtol=1e-3, atol=1e-6, err_msg=name)


def test_resident_groups_are_independent_launches(monkeypatch):
    """A handle whose sub-nets do not fit the register file at once trains them in groups, one epoch launch per group: the
    ten sub-nets of configs[1] as 2 x 5 must give, bit for bit, what two handles of five sub-nets (global indices 0-4 and 5-9)
    give on their own."""
    monkeypatch.setenv("DIMN_RESIDENT", "1")
    cfg, norm, targets, preds, train, val = _cfg2_problem()

    def run(k0, k1):
        e = _hip()([len(preds[k]) for k in range(k0, k1)], cfg["H"], cfg["O"], batch_size=cfg["B"], dropout_rate=0.2, learning_rate=1e-3,
                   seed=1234, subnet_offset=k0)
        e.set_matrix(norm)
        for i, k in enumerate(range(k0, k1)):
            e.set_indices(i, preds[k], targets[k])
        e.gather(True)
        e.set_split(train[:5 * 64 + 9], val)
        e.init_weights()
        e.set_profiling(True)
        out = [e.train_epoch(0), e.train_epoch(1), e.val_loss(), [e.get_weights(i) for i in range(k1 - k0)]]
        assert e.get_timers()[7] == e.step_count() == 12
        e.close()
        return out

    whole, lo, hi = run(0, 10), run(0, 5), run(5, 10)
    for j in range(3):
        assert np.array_equal(whole[j], np.concatenate([lo[j], hi[j]]))
    for i in range(10):
        for x, y in zip(whole[3][i], (lo[3] + hi[3])[i]):
            assert np.array_equal(x, y)


def test_resident_launch_that_aborts_is_undone_and_rerun_on_the_streaming_kernels(monkeypatch, capfd):
    """A resident epoch launch that times out (a GPU shared with another process: DIMN_RES_TEST="abort=2" makes the host treat
    the launch of epoch 1 as timed out) must not fail the fit: the pre-epoch state is restored, the epoch re-runs on the
    streaming kernels and the handle stays on them -- all three epochs still match the oracle."""
    from helpers import make_problem, load_problem
    monkeypatch.setenv("DIMN_RESIDENT", "1")
    monkeypatch.setenv("DIMN_RES_TEST", "abort=2")
    prob = make_problem(n=500, g=900, Ds=[300, 280], H=256, O=512, seed=5, val_frac=0.1)
    kw = dict(batch_size=64, dropout_rate=0.2, learning_rate=1e-3, seed=99)
    a, b = load_problem(_hip(), prob, **kw), load_problem(_oracle(), prob, **kw)
    a.init_weights(); b.init_weights()
    a.set_profiling(True)
    ran_resident = []
    for epoch in range(3):
        np.testing.assert_allclose(a.train_epoch(epoch), b.train_epoch(epoch), rtol=1e-4)
        ran_resident.append(a.get_timers()[7] > 0)
    assert ran_resident == [True, True, False]          # epoch 1 launched (and was undone), epoch 2 never tried
    assert "re-runs on the streaming kernels" in capfd.readouterr().err
    np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=1e-4)
    np.testing.assert_allclose(a.predict(), b.predict(), rtol=1e-4, atol=1e-6)
    a.close(); b.close()


def test_resident_hand_off_retry_path():
    """The sentinel hand-off of the register-resident kernel under stress: libdimn_nocanary.so is the same library without
    the canary polls, so the bulk requests of every hand-off leave before most of the data has landed and res_fix's
    re-request loop -- which a normal run almost never enters -- carries the protocol.  The R variants of the parity
    tests and the 8-GPU-share test must pass unchanged."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(os.path.dirname(here), "deepimpute_amd", "csrc", "libdimn_nocanary.so")
    if not os.path.exists(lib):
        pytest.skip("libdimn_nocanary.so not built (make -C deepimpute_amd/csrc)")
    env = dict(os.environ, DIMN_LIB_PATH=lib)
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), os.path.join(here, "test_gpu_configs.py"), "-q", "-x",
                          "-m", "gpu", "-k", "(second_layer_paths and R) or resident_epoch_kernel"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert " passed" in res.stdout and "failed" not in res.stdout


@pytest.mark.parametrize("K,precision,want", [
    (5, "fp32", dict(path="resident", resident_groups=1, resident_splits=3)),        # one rank of the 8-GPU job
    (10, "fp32", dict(path="resident", resident_groups=2)),                          # 4-GPU share: two groups of 5
    (20, "fp32", dict(path="resident", resident_groups=4, resident_splits=3)),       # 2-GPU share: four groups of 5
    (16, "fp32", dict(path="streaming", mid_fused=0, first_layer=1)),                 # four groups of 4 would only draw: the streaming kernels (two-kernel second layer)
    (40, "fp32", dict(path="streaming", mid_fused=1, mid_slices=6, mid_keep=2, train_bf16=0, first_layer=1)),   # the single-GPU job (mid_keep 2: the tile pipeline k_mid_pipe)
    (40, "bf16", dict(path="streaming", mid_fused=1, mid_slices=6, mid_keep=2, train_bf16=1)),          # bf16 operands: k_mid_pipe<BF>
])
def test_automatic_path_choice(K, precision, want, monkeypatch):
    """The kernels dimn_create picks BY ITSELF (no DIMN_* variable set) for the sub-net counts a rank of the 50k x 20k job
    sees at 8 / 4 / 2 / 1 GPUs -- DESIGN.md's decision table, asserted through dimn_path_info, and that a short epoch on
    that path really runs it (timer slot [7] counts the steps of resident launches)."""
    import bench
    for name in list(os.environ):
        if name.startswith("DIMN_") and name not in ("DIMN_LIB_PATH", "DIMN_HOST_THREADS"):
            monkeypatch.delenv(name)
    cfg = dict(bench.CONFIGS["cfg3"])
    targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=0)
    norm = bench.synth_counts(256, cfg["g"], seed=0)
    ks = list(range(K))
    train, val = np.arange(0, 64 + 30, dtype=np.int32), np.arange(200, 256, dtype=np.int32)
    kw = dict(batch_size=64, dropout_rate=0.2, learning_rate=1e-4, seed=1234)
    if precision != "fp32":
        kw["precision"] = precision
    e = _load(_hip(), cfg, norm, preds, targets, ks, train, val, **kw)
    info = e.path_info()
    for key, value in want.items():
        assert info[key] == value, (key, info)
    assert e.training_precision == ("bf16" if info["train_bf16"] else "fp32")
    e.set_profiling(True)
    assert np.isfinite(e.train_epoch(0)).all()
    t = e.get_timers()
    assert (t[7] == 2) == (info["path"] == "resident"), (t, info)
    e.close()
