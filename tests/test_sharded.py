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


def test_shard_subnets_balances_by_predictor_count():
    """ranks are balanced by sum of D_k (a rank's step time and arena follow it), in contiguous blocks; near-equal D_k
    -- the 50k x 20k job, configs[4]'s 59 sub-nets -- keep the by-count split."""
    assert shard_subnets(40, 8, [2390 + (7 * k) % 30 for k in range(40)])[0] == [5] * 8
    assert shard_subnets(59, 8, [2400 + (k % 7) for k in range(59)])[0] == [8, 8, 8, 7, 7, 7, 7, 7]
    # a user-supplied gene list with two heavy sub-nets at the end: by count [3, 3, 2, 2] would give the last rank 1800
    counts, offs = shard_subnets(10, 4, [100] * 8 + [900, 900])
    assert counts == [4, 4, 1, 1] and offs == [0, 4, 8, 9]
    w = [50, 700, 60, 40, 30, 650, 45, 55, 35]
    counts, offs = shard_subnets(9, 3, w)
    loads = [sum(w[offs[r]:offs[r] + counts[r]]) for r in range(3)]
    assert sum(counts) == 9 and min(counts) >= 1
    best = min(max(sum(w[:i]), sum(w[i:j]), sum(w[j:])) for i in range(1, 8) for j in range(i + 1, 9))
    assert max(loads) == best                                  # the optimum of the contiguous partitions
    assert shard_subnets(5, 5, [1, 2, 3, 4, 5])[0] == [1] * 5   # every rank keeps one sub-net
    assert shard_subnets(4, 6)[0] == [1, 1, 1, 1, 0, 0]         # (more ranks than sub-nets: the caller refuses)
    with pytest.raises(ValueError):
        shard_subnets(3, 2, [1.0, 0.0, 2.0])


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
