"""N > 1 on hardware: one process per GPU, sub-nets sharded, RCCL over xGMI (BASELINE configs[3]).  Needs >= 2
visible GPUs (skipped on a 1-GPU box, where the same worker still runs as a 1-rank RCCL job).  A job of N ranks
must reproduce the 1-rank job: same early-stopping epoch, losses and imputed values to fp32 rounding (the split-K
partition of the first layer depends on how many sub-nets share a GPU, so not bit for bit -- DESIGN.md section 4)."""
import json
import os
import signal
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _ndev():
    from deepimpute_amd import _lib
    return _lib.device_count()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_job(world, out, timeout=900, mode=""):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", DIMN_HOST_THREADS="4", DIMN_RCCL_WORKER_MODE=mode)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True))
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:                                    # a hung rank must not outlive the test (exact pids only)
            if p.poll() is None:
                os.killpg(p.pid, signal.SIGKILL)
    for rank, (p, log) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, "rank %d of %d:\n%s" % (rank, world, log[-4000:])
    return np.load(out)


@pytest.fixture(scope="module")
def single(tmp_path_factory):
    return _run_job(1, str(tmp_path_factory.mktemp("rccl") / "w1.npz"))


def test_single_rank_rccl_job_runs(single):
    assert int(single["K"]) == 9 and int(single["epochs"]) >= 1
    assert np.isfinite(single["imputed"]).all() and np.isfinite(single["full"]).all()
    np.testing.assert_allclose(single["reloaded"], single["imputed"], rtol=1e-6)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_sharded_job_equals_single_rank(tmp_path, single, world):
    if _ndev() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, _ndev()))
    multi = _run_job(world, str(tmp_path / ("w%d.npz" % world)))
    assert int(multi["epochs"]) == int(single["epochs"])
    np.testing.assert_allclose(multi["val"], single["val"], rtol=2e-5)
    np.testing.assert_allclose(multi["loss"], single["loss"], rtol=2e-5)
    np.testing.assert_allclose(multi["imputed"], single["imputed"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(multi["reloaded"], multi["imputed"], rtol=1e-6)
    np.testing.assert_allclose(multi["metrics"], single["metrics"], rtol=1e-4)
    np.testing.assert_allclose(multi["vsum"], single["vsum"], rtol=2e-5)
    np.testing.assert_allclose(multi["full"], single["full"], rtol=1e-4, atol=1e-6)


def test_rccl_gather_at_cfg3_share_shapes(tmp_path):
    """The first multi-GPU box also checks the big-shape gather: 40 sub-nets of the 50k x 20k job, 20 per rank at world 2
    ([n][20 * 512] blocks placed side by side in root's [n][40 * 512] matrix), against the same job in one process."""
    one = _run_job(1, str(tmp_path / "big1.npz"), mode="big")
    assert one["full"].shape[1] == 40 * 512 and np.isfinite(one["full"]).all()
    if _ndev() < 2:
        pytest.skip("needs 2 GPUs, %d visible (the 1-rank leg ran)" % _ndev())
    two = _run_job(2, str(tmp_path / "big2.npz"), mode="big")
    np.testing.assert_allclose(two["vsum"], one["vsum"], rtol=2e-5)
    np.testing.assert_allclose(two["full"], one["full"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("world", [1, 2])
def test_bench_contract_line(tmp_path, world):
    """`bench.py --gpus N` as the driver launches it prints exactly one JSON line with the contract's keys."""
    if _ndev() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, _ndev()))
    args = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "0", "--config", "tiny", "--epochs", "2", "--no-cpu-baseline"]
    if world == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    p = subprocess.Popen(cmd, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=900)
    finally:
        if p.poll() is None:
            os.killpg(p.pid, signal.SIGKILL)
    assert p.returncode == 0, err[-4000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["n_gpus"] == world and rec["value"] > 0 and rec["roofline"]["achieved"] > 0
    if world > 1:
        pr = rec["config"]["per_rank"]
        assert rec["config"]["collectives"] == "rccl" and pr["nranks_ncclCommCount"] == world
        assert len(pr["lane_step_ms"]) == world and all(x > 0 for x in pr["lane_step_ms"]) and pr["gather_bytes_into_root"] > 0
