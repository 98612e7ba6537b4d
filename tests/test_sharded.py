"""N>1 path on CPU: world_size-2 (and 3) gloo jobs must reproduce the single-process result
exactly -- same early-stopping epoch, same gathered predictions (Philox keys use the global
sub-net index, so sharding cannot change the numbers)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from deepimpute_amd.sharded import shard_subnets

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, out, mode=""):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2", SHARDED_WORKER_MODE=mode)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return np.load(out)


def test_shard_subnets():
    assert shard_subnets(40, 8) == ([5] * 8, list(range(0, 40, 5)))
    assert shard_subnets(10, 4) == ([3, 3, 2, 2], [0, 3, 6, 8])
    assert shard_subnets(3, 1) == ([3], [0])


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_equals_single_process(tmp_path, world):
    single = _run(1, str(tmp_path / "single.npz"))
    multi = _run(world, str(tmp_path / ("w%d.npz" % world)))
    assert int(single["K"]) >= world
    assert int(single["epochs"]) == int(multi["epochs"])
    np.testing.assert_allclose(multi["val"], single["val"], rtol=1e-12)
    np.testing.assert_allclose(multi["loss"], single["loss"], rtol=1e-12)
    assert np.array_equal(multi["imputed"], single["imputed"])
    np.testing.assert_allclose(multi["metrics"], single["metrics"], rtol=1e-12)


def test_sharded_fit_without_output_prefix_shares_one_directory(tmp_path):
    """MultiNet's default output_prefix is a per-process temporary directory: a sharded job must still put every rank's
    shard into ONE directory (named after the job), assemble them on rank 0 and let a fresh object load them."""
    multi = _run(2, str(tmp_path / "dd.npz"), mode="default_dir")
    single = _run(1, str(tmp_path / "single.npz"))
    assert np.array_equal(multi["imputed"], single["imputed"])


def test_refit_with_fewer_ranks_removes_stale_shards(tmp_path):
    """Shards of ranks the current job does not have (an older fit with more ranks into the same directory) must not
    survive: their global sub-net indices overlap the fresh shards'."""
    out = str(tmp_path / "w2.npz")
    os.makedirs(out + ".dir")
    for r in (2, 10):
        np.savez(os.path.join(out + ".dir", "model.rank%d.npz" % r), W1_0=np.full((3, 3), 7.0, np.float32))
    _run(2, out)
    left = sorted(f for f in os.listdir(out + ".dir") if f.startswith("model.rank"))
    assert left == ["model.rank0.npz", "model.rank1.npz"], left
