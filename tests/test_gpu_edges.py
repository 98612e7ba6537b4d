"""Edge cases of the hot path on the GPU, each against the oracle: degenerate sizes (one predictor, one cell per split,
fewer training cells than a batch, one sub-net), empty requests, and the same through the register-resident, general and
bf16 paths."""
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


def _compare(prob, epochs=2, rtol=1e-4, **kw):
    a, b = load_problem(_hip(), prob, **kw), load_problem(_oracle(), prob, **kw)
    a.init_weights(); b.init_weights()
    for e in range(epochs):
        np.testing.assert_allclose(a.train_epoch(e), b.train_epoch(e), rtol=rtol)
        np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=rtol)
    assert a.step_count() == b.step_count()
    np.testing.assert_allclose(a.predict(), b.predict(), rtol=rtol, atol=1e-6)
    return a, b


@pytest.mark.parametrize("resident", ["0", "1"])
@pytest.mark.parametrize("n_train,n_val,B", [(5, 1, 64), (64, 3, 64), (65, 70, 64), (1, 1, 1), (130, 9, 7)])
def test_tiny_splits_and_partial_batches(n_train, n_val, B, resident, monkeypatch):
    """Fewer training cells than a batch, exactly one batch, one cell over, a single cell, an odd small batch size --
    on the streaming kernels and (H = 256) on the register-resident epoch kernel."""
    monkeypatch.setenv("DIMN_RESIDENT", resident)
    prob = make_problem(n=n_train + n_val + 5, g=120, Ds=[40, 17], H=256, O=32, seed=2)
    prob["train"] = np.arange(n_train, dtype=np.int32)
    prob["val"] = np.arange(n_train, n_train + n_val, dtype=np.int32)
    a, b = _compare(prob, batch_size=B, dropout_rate=0.25, learning_rate=1e-3, seed=9)
    a.close(); b.close()


def test_single_predictor_single_subnet_and_ragged_widths():
    prob = make_problem(n=150, g=60, Ds=[1], H=24, O=5, seed=4)              # D = 1, O = 5, H = 24: everything padded
    a, b = _compare(prob, batch_size=32, dropout_rate=0.0, learning_rate=2e-3, seed=1)
    a.close(); b.close()
    prob = make_problem(n=150, g=400, Ds=[16, 17, 15, 300], H=384, O=17, seed=5)     # the widest tuned hidden layer, D around a chunk edge
    a, b = _compare(prob, batch_size=64, dropout_rate=0.5, learning_rate=1e-3, seed=2)
    a.close(); b.close()


def test_empty_and_repeated_requests():
    prob = make_problem(n=90, g=80, Ds=[20], H=32, O=16, seed=6)
    a = load_problem(_hip(), prob, batch_size=16, learning_rate=1e-3, seed=3)
    a.init_weights()
    assert a.predict(np.zeros(0, np.int32)).shape == (0, 16)
    rows = np.array([5, 5, 5, 0, 89], np.int32)                           # repeated rows
    out = a.predict(rows)
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[1], out[2])
    full = a.predict()
    assert np.array_equal(out[3], full[0]) and np.array_equal(out[4], full[89])
    first = a.train_epoch(0)
    a.init_weights()                                                      # re-initialising replays the run exactly
    assert np.array_equal(a.train_epoch(0), first)
    a.close()


def test_general_and_bf16_paths_on_degenerate_sizes():
    from deepimpute_amd.engine import HipGeneralEngine
    from oracle.dimo import GeneralOracleEngine
    prob = make_problem(n=40, g=50, Ds=[3, 1], H=8, O=2, seed=7)
    prob["train"], prob["val"] = np.arange(0, 33, dtype=np.int32), np.arange(33, 34, dtype=np.int32)
    engines = []
    for cls in (HipGeneralEngine, GeneralOracleEngine):
        e = cls(prob["Ds"], [(8, "tanh", 0.5), (3, "relu", 0.0)], 2, batch_size=100, learning_rate=1e-3, seed=5, loss="mae")
        e.set_matrix(prob["norm"])
        for k in range(2):
            e.set_indices(k, prob["pred"][k], prob["targ"][k])
        e.gather(True)
        e.set_split(prob["train"], prob["val"])
        e.init_weights()
        engines.append(e)
    a, b = engines
    for epoch in range(3):
        np.testing.assert_allclose(a.train_epoch(epoch), b.train_epoch(epoch), rtol=1e-4)
    np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=1e-4)
    np.testing.assert_allclose(a.predict(), b.predict(), rtol=1e-4, atol=1e-6)
    a.close(); b.close()
    prob = make_problem(n=70, g=90, Ds=[5, 33], H=256, O=20, seed=8)
    prob["train"], prob["val"] = np.arange(0, 3, dtype=np.int32), np.arange(3, 70, dtype=np.int32)
    a = load_problem(_hip(), prob, batch_size=64, dropout_rate=0.2, learning_rate=1e-3, seed=5, precision="bf16")
    mode = a.path_info()["train_bf16"]               # which training GEMMs take bf16 operands on the path the library picked (2: all, resident kernel)
    b = load_problem(_oracle(), prob, batch_size=64, dropout_rate=0.2, learning_rate=1e-3, seed=5, precision="bf16", infer_bf16=True, train_bf16=mode)
    a.init_weights(); b.init_weights()
    np.testing.assert_allclose(a.train_epoch(0), b.train_epoch(0), rtol=1e-3 if mode else 1e-4)
    np.testing.assert_allclose(a.val_loss(), b.val_loss(), rtol=1e-3 if mode else 5e-4)
    np.testing.assert_allclose(a.predict(), b.predict(), rtol=5e-3 if mode else 2e-3, atol=5e-4 if mode else 2e-4)
    a.close(); b.close()


def test_loopback_gather_places_blocks_like_hstack():
    """dimn_comm_gather_loopback: the root-side code of dimn_comm_gather_predictions (arena sizing, block offsets, the strided
    placement into [n][K_global * O]) with device-to-device copies where ncclRecv would run -- 59 sub-nets over 8 "ranks"
    (8 + 8 + 8 + 7 + 7 + 7 + 7 + 7: configs[4]'s split) on one GPU, against np.hstack of the per-rank predictions; then
    dimn_impute_finish(from_gathered) over the gathered matrix against the same epilogue over one handle that owns all 59."""
    from deepimpute_amd.sharded import shard_subnets
    HipEngine = _hip()
    K, world, n, g, H, O = 59, 8, 333, 400, 32, 16
    rng = np.random.default_rng(11)
    prob = make_problem(n=n, g=g, Ds=[int(d) for d in rng.integers(9, 40, size=K)], H=H, O=O, seed=8)
    counts, offs = shard_subnets(K, world)
    assert counts == [8, 8, 8, 7, 7, 7, 7, 7]
    whole = load_problem(HipEngine, prob, batch_size=32, learning_rate=1e-3, seed=21)
    whole.init_weights()
    engines = []
    for r in range(world):
        ks = range(offs[r], offs[r] + counts[r])
        e = HipEngine([prob["Ds"][k] for k in ks], H, O, batch_size=32, learning_rate=1e-3, seed=21, subnet_offset=offs[r])
        e.set_matrix(prob["norm"])
        for i, k in enumerate(ks):
            e.set_indices(i, prob["pred"][k], prob["targ"][k])
        e.gather(True)
        e.set_split(prob["train"], prob["val"])
        e.init_weights()                                        # keyed by the GLOBAL sub-net index: the same weights as `whole`
        engines.append(e)
    full = whole.predict()
    for root in (0, 3):
        for e in engines:
            e.predict_device()
        got = HipEngine.gather_loopback(engines, n, root=root)
        assert got.shape == (n, K * O)
        np.testing.assert_array_equal(got, np.hstack([e.predict() for e in engines]))
        np.testing.assert_allclose(got, full, rtol=1e-5, atol=1e-6)      # (split-K partition differs with the sub-nets per handle)
    # the epilogue of predict() over root's gathered matrix == over one handle's own predictions
    raw = np.rint(np.expm1(prob["norm"].astype(np.float64)))
    slots = np.concatenate([prob["targ"][k] for k in range(K)])
    order = np.lexsort((np.arange(slots.size), slots))
    gene_off = np.zeros(g + 1, np.int64)
    np.cumsum(np.bincount(slots, minlength=g), out=gene_off[1:])
    ceiling = 2 * np.log1p(raw.max())
    for e in engines:
        e.predict_device()
    HipEngine.gather_loopback(engines, n, root=0, want_host=False)
    a = engines[0].impute_finish(raw, gene_off, order, "restore", ceiling, from_gathered=True)
    whole.predict_device()
    b = whole.impute_finish(raw, gene_off, order, "restore", ceiling)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    for e in engines + [whole]:
        e.close()


def test_arena_cache_hands_blocks_back_and_can_be_emptied(monkeypatch):
    """Large device blocks outlive their handle inside the process (include/dimn.h: dimn_release_cached_memory): a second engine of
    the same shapes trains to the same bits on recycled arenas, with the cache emptied in between, and with the cache off."""
    from deepimpute_amd import _lib
    prob = make_problem(n=9000, g=1500, Ds=[700, 650], H=64, O=32, seed=12)      # X arena 2 x 9000 x ~700 x 4 B = 50 MB: above the 32 MB threshold

    def run():
        e = load_problem(_hip(), prob, batch_size=64, learning_rate=1e-3, seed=4)
        e.init_weights()
        loss = e.train_epoch(0)
        out = e.predict(np.arange(50, dtype=np.int32))
        e.close()
        return loss, out
    first = run()
    again = run()                                                # matrix and arenas come from the cache (stale contents of the first life)
    assert _lib.load()["release_cached_memory"]() == 0
    fresh = run()
    monkeypatch.setenv("DIMN_ARENA_CACHE_GB", "0")
    uncached = run()
    for other in (again, fresh, uncached):
        np.testing.assert_array_equal(first[0], other[0])
        np.testing.assert_array_equal(first[1], other[1])


def test_sharded_fit_and_gather_on_hip_engines_as_threads_of_one_gpu():
    """The N > 1 leg of configs[3] as far as one GPU can carry it: 8 HIP handles, one per "rank" (threads), run sharded.fit_sharded --
    one all-reduce of the two loss sums per epoch, the GLOBAL early-stopping decision of multinet.py:242-243 -- and sharded.predict_sharded
    with the device gather (dimn_comm_gather_loopback: everything of the RCCL gather but RCCL).  Against ONE handle that owns all the
    sub-nets: the same stopping epoch, the same loss curves and predictions to fp32 rounding (the first layer's split-K partition
    depends on how many sub-nets share a handle), and against the oracle at the usual tolerance."""
    import threading
    from deepimpute_amd.sharded import fit_sharded, predict_sharded, shard_subnets
    from loopback_comm import LoopbackComm, LoopbackWorld
    HipEngine = _hip()
    K, world, H, O = 19, 8, 48, 32
    rng = np.random.default_rng(5)
    prob = make_problem(n=700, g=500, Ds=[int(d) for d in rng.integers(20, 90, size=K)], H=H, O=O, seed=14)
    kw = dict(batch_size=64, dropout_rate=0.2, learning_rate=2e-3, seed=31)
    whole = load_problem(HipEngine, prob, **kw)
    whole.init_weights()
    ref_epochs, ref_loss, ref_val = whole.fit(40, 2)
    ref_pred = whole.predict()
    assert 2 < ref_epochs < 40                                 # early stopping decides, not the epoch limit
    counts, offs = shard_subnets(K, world, weights=prob["Ds"])
    assert sum(counts) == K and min(counts) >= 1
    shared = LoopbackWorld(world)
    engines, results, errors = [], [None] * world, []
    for r in range(world):
        ks = range(offs[r], offs[r] + counts[r])
        e = HipEngine([prob["Ds"][k] for k in ks], H, O, subnet_offset=offs[r], **kw)
        e.set_matrix(prob["norm"])
        for i, k in enumerate(ks):
            e.set_indices(i, prob["pred"][k], prob["targ"][k])
        e.gather(True)
        e.set_split(prob["train"], prob["val"])
        e.init_weights()
        engines.append(e)

    def rank_main(r):
        try:
            comm = LoopbackComm(shared, r)
            epochs, loss, val = fit_sharded(engines[r], comm, 40, 2)
            block = predict_sharded(engines[r], comm, counts)
            results[r] = (epochs, loss, val, block)
        except Exception as exc:                                # a dead rank would leave the others at the barrier
            errors.append((r, exc))
            shared.barrier.abort()
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    for r in range(world):
        epochs, loss, val, block = results[r]
        assert epochs == ref_epochs                              # every rank took the single-process stop decision
        np.testing.assert_allclose(loss, ref_loss, rtol=2e-5)
        np.testing.assert_allclose(val, ref_val, rtol=2e-5)
        assert (block is not None) == (r == 0)
    np.testing.assert_allclose(results[0][3], ref_pred, rtol=2e-4, atol=2e-6)
    ora = load_problem(_oracle(), prob, **kw)
    ora.init_weights()
    o_epochs, o_loss, o_val = ora.fit(40, 2)
    assert o_epochs == ref_epochs
    np.testing.assert_allclose(results[0][3], ora.predict(), rtol=1e-4, atol=1e-6)
    for e in engines + [whole, ora]:
        e.close()


def test_multinet_close_gives_the_cached_device_blocks_back(tmp_path):
    """MultiNet.close() is documented as releasing the GPU; since the process-wide block cache (round 4) it must also empty
    that cache (dimn_release_cached_memory) -- close(release_cache=False) keeps it for a fit() that follows -- and a constructor touches
    nothing.  Observed through dimn_cached_memory_info (ABI 8)."""
    import pandas as pd
    from deepimpute_amd import _lib, release_cached_memory
    from deepimpute_amd.multinet import MultiNet
    release_cached_memory()
    assert _lib.cached_memory_info() == (0, 0)
    rng = np.random.default_rng(3)
    n, g = 9000, 1600                                            # counts 58 MB, X arena > 32 MB: cached-class blocks
    raw = pd.DataFrame(rng.poisson(rng.gamma(2.0, 2.0, size=g), size=(n, g)).astype(np.float64), index=["c%d" % i for i in range(n)],
                       columns=["g%d" % j for j in range(g)])
    net = MultiNet(output_prefix=str(tmp_path), sub_outputdim=256, verbose=0, max_epochs=1, seed=3)
    assert _lib.cached_memory_info() == (0, 0)                   # constructing allocates nothing
    net.fit(raw, NN_lim=g)
    idle, owned = _lib.cached_memory_info()
    assert owned > (32 << 20)
    net.close(release_cache=False)
    idle2, owned2 = _lib.cached_memory_info()
    assert owned2 == 0 and idle2 >= owned                        # everything the handle and the counts owned waits in the cache
    net.close()
    assert _lib.cached_memory_info() == (0, 0)
