"""CPU checks of bench.py's workload generator and sharding helpers (the GPU legs are exercised by the
driver; these make sure every rank builds the same problem and that the shards cover it)."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_synthetic_matrix_is_a_pure_function_of_its_seed():
    a = bench.synth_counts(700, 90, seed=3, threads=1)
    b = bench.synth_counts(700, 90, seed=3, threads=5)          # every rank may use a different pool size
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert not np.array_equal(a, bench.synth_counts(700, 90, seed=4, threads=2))
    counts = np.expm1(a.astype(np.float64))
    assert np.allclose(counts, np.round(counts), atol=1e-3) and counts.max() >= 10     # raw counts, passes inspect_data


def test_index_lists_have_the_reference_shapes():
    g, O = 1300, 128
    targets, preds = bench.synth_indices(g, O, seed=0)
    K = -(-g // O)
    assert targets.shape == (K, O) and len(preds) == K
    assert set(targets[:g // O].ravel()) <= set(range(g))
    assert np.unique(targets.ravel()[:g]).size == g             # every gene is a target once before the random fill
    for k in range(K):
        assert np.unique(preds[k]).size == preds[k].size        # first-occurrence unique, like setPredictors
        assert not set(preds[k]) & set(targets[k])              # predictors exclude the sub-net's own targets
    t2, p2 = bench.synth_indices(g, O, seed=0)
    assert np.array_equal(targets, t2) and all(np.array_equal(a, b) for a, b in zip(preds, p2))


def test_split_and_shards_cover_everything():
    train, val = bench.split_rows(1000, seed=0)
    assert val.size == 50 and np.array_equal(np.sort(np.concatenate([train, val])), np.arange(1000))
    assert np.array_equal(train, np.sort(train))               # np.setdiff1d order, multinet.py:229
    for K, world in ((40, 1), (40, 8), (10, 4), (7, 3)):
        counts, offs = bench.shard(K, world)
        assert sum(counts) == K and offs[0] == 0 and max(counts) - min(counts) <= 1
        assert all(offs[r + 1] == offs[r] + counts[r] for r in range(world - 1))


def test_file_rendezvous_hands_the_payload_to_every_rank(monkeypatch, tmp_path):
    monkeypatch.setenv("MASTER_PORT", "4%d" % (os.getpid() % 10000))
    ranks = [bench.FileRendezvous(r, 3) for r in range(3)]
    got = {}

    def run(r):
        got[r] = ranks[r].broadcast_bytes("uid", b"\x01\x02payload" if r == 0 else None, timeout=20.0)
    th = [threading.Thread(target=run, args=(r,)) for r in (2, 1, 0)]     # root arrives last
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == {0: b"\x01\x02payload", 1: b"\x01\x02payload", 2: b"\x01\x02payload"}
    ranks[0].cleanup()
    assert not os.path.exists(ranks[0].dir)


def test_rccl_failure_yields_a_null_value_line_on_every_rank(monkeypatch):
    """An N > 1 run whose RCCL communicator does not come up must not look like a measurement: every rank learns of the
    failure (file vote), and the line rank 0 prints has value null and names the error."""
    import argparse
    monkeypatch.setenv("MASTER_PORT", "5%d" % (os.getpid() % 10000))

    class FakeEngine:
        def __init__(self, rank, broken):
            self.rank, self.broken = rank, broken

        def comm_unique_id(self):
            return np.arange(128, dtype=np.uint8)

        def comm_init(self, uid, world, rank):
            assert np.array_equal(uid, np.arange(128, dtype=np.uint8))
            if self.broken:
                raise RuntimeError("ncclCommInitRank failed: unhandled system error")

        def comm_allreduce_sum(self, v):
            return v

        def comm_info(self):
            return 3, self.rank

    for broken_rank in (None, 1):
        rd = [bench.FileRendezvous(r, 3) for r in range(3)]
        got = {}

        def run(r):
            got[r] = bench.bring_up_rccl(FakeEngine(r, r == broken_rank), rd[r], r, 3)
        th = [threading.Thread(target=run, args=(r,)) for r in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        rd[0].cleanup()
        if broken_rank is None:
            assert all(isinstance(got[r][0], bench.RcclBenchComm) and got[r][1] is None for r in range(3))
        else:
            assert all(got[r][0] is None and got[r][1] for r in range(3))
            assert "ncclCommInitRank" in got[1][1] and "other rank" in got[0][1]
    # a rank whose ENGINE cannot be built (dimn_create on a missing device: --gpus 2 on a one-GPU box) ends the same way, and no
    # rank reaches ncclCommInitRank (FakeEngine.comm_init would raise AssertionError through `reached`)
    reached = []

    class Untouched(FakeEngine):
        def comm_unique_id(self):
            reached.append(self.rank)
            return super().comm_unique_id()

    def factory(r):
        if r == 2:
            raise RuntimeError("libdimn error -1: dimn_create: device_id 2 out of range (1 devices)")
        return Untouched(r, False)
    rd = [bench.FileRendezvous(r, 3) for r in range(3)]
    got = {}

    def run_job(r):
        got[r] = bench.bring_up_job(lambda: factory(r), rd[r], r, 3)
    th = [threading.Thread(target=run_job, args=(r,)) for r in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    rd[0].cleanup()
    assert not reached
    assert all(got[r][1] is None and got[r][2] for r in range(3))
    assert got[2][0] is None and "device_id 2 out of range" in got[2][2] and "other rank" in got[0][2] and got[0][0] is not None
    args = argparse.Namespace(steps=2, warmup=1, precision="fp32")
    line = bench.rccl_failure_line(args, 8, "50k x 20k", "boom")
    assert line["value"] is None and line["ms_per_step"] is None and line["rccl_error"] == "boom" and line["n_gpus"] == 8
    for key in ("metric", "unit", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line
    assert not hasattr(bench, "FileBenchComm")          # no transport other than RCCL can produce a number


def test_deviation_reports_the_tolerance_of_the_parity_tests():
    b = np.array([1.0, 2.0, 1e-3, 4.0])
    a = b * np.array([1.0, 1.0 + 5e-5, 1.0, 1.0 + 3e-4])
    d = bench.deviation(a, b)
    assert abs(d["max_rel"] - 3e-4) < 1e-9 and d["elements"] == 4
    assert abs(d["outside_tolerance"] - 0.25) < 1e-12 and abs(d["max_abs"] - 1.2e-3) < 1e-9      # only the last element misses 1e-4 |b| + 1e-5


def test_accuracy_leg_compares_element_wise_at_the_full_horizon(monkeypatch):
    """bench.accuracy_pair's plumbing on the CPU: the oracle stands in for the HIP engine (so `hip` must equal `cpu_port` to the bit,
    and float32 vs float64 must differ a little): all K sub-nets on the `hip` side, the first n_subnets on the ports, the element-wise
    deviation over every sample cell."""
    from deepimpute_amd import engine
    from oracle.dimo import OracleEngine

    class Standin(OracleEngine):
        def path_info(self):
            return {"path": "oracle stand-in"}
    monkeypatch.setattr(engine, "HipEngine", Standin)
    cfg = {"H": 24, "O": 32, "B": 16, "n": 240, "g": 200}
    norm = bench.synth_counts(cfg["n"], cfg["g"], seed=1, threads=2)
    targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=0)
    acc = bench.accuracy_pair(cfg, targets, preds, norm, 3, 1e-3, n_cells=200, n_subnets=3)
    iv = acc["imputed_values"]["log1p_space"]
    assert acc["hip"]["subnets_trained"] == targets.shape[0] and acc["cpu_port"]["subnets_trained"] == 3
    assert iv["hip_vs_cpu_port"]["max_abs"] == 0.0                                        # global Philox keys: K does not change a sub-net
    assert iv["hip_share_vs_cpu_port_fp64"] == iv["hip_vs_cpu_port_fp64"] == iv["cpu_port_fp32_vs_fp64"]
    assert set(acc["imputed_values"]["by_epoch"]) == {"1", "3"} and acc["imputed_values"]["within_noise_floor"] is True
    assert len(acc["imputed_values"]["rms_rel_per_subnet"]) == 3
    assert iv["hip_vs_cpu_port"]["elements"] == 200 * 3 * 32
    assert 0 < iv["cpu_port_fp32_vs_fp64"]["max_rel"] < 1e-3 and iv["cpu_port_fp32_vs_fp64"]["outside_tolerance"] == 0.0
    assert acc["relative_difference"]["val_loss"] == 0.0
    assert acc["imputed_values"]["counts_expm1"]["hip_vs_cpu_port"]["max_rel"] == 0.0
