"""One host copy of the matrix for the ranks of a node (deepimpute_amd._shm; VERDICT r05 item 2, SURVEY 8e "cfg5 streams row blocks
from pinned host to all GPUs" -- one source, not eight): a world-2 gloo job on CPU whose frame exists once, against the same job
run by one process on a private frame."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_share_one_copy_of_the_frame_and_of_the_log1p_matrix(tmp_path):
    out = str(tmp_path / "shm.npz")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "shm_worker.py"), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    r0, r1 = (json.load(open("%s.rank%d.json" % (out, r))) for r in range(2))
    # ONE physical copy: both ranks mapped the same inode of /dev/shm ...
    assert r0["identity"] == r1["identity"] and r0["norm_identity"] == r1["norm_identity"] and r0["identity"] != r0["norm_identity"]
    size_kb = r0["frame_bytes"] // 1024
    for rep in (r0, r1):
        fs = rep["frame_smaps"]
        assert fs["Size"] >= size_kb
        # ... every page of it resident in both, and charged HALF to each (Pss = resident pages / sharers): shared pages, not two copies
        assert fs["Rss"] >= 0.95 * size_kb and fs["Pss"] <= 0.55 * fs["Rss"], fs
        assert rep["norm_equal"]
    # rank 1 never owned a private page of the frame (it maps it read-only)
    assert r1["frame_smaps"]["Private_Dirty"] == 0 and r1["frame_smaps"].get("Anonymous", 0) == 0
    # ... and the job computes what one process on a private frame computes
    from helpers import multinet_with
    from oracle.dimo import OracleEngine
    rng = np.random.default_rng(3)
    n, g = 4096, 2048
    mu = rng.lognormal(0.5, 1.2, size=g)
    raw = pd.DataFrame(rng.poisson(rng.gamma(2.0, mu / 2.0, size=(n, g))).astype(np.float64),
                       index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
    net = multinet_with(OracleEngine, max_epochs=2, patience=2, learning_rate=2e-3, seed=17, sub_outputdim=64, ncores=1, verbose=0,
                        output_prefix=str(tmp_path / "single"),
                        architecture=[{"type": "dense", "neurons": 32, "activation": "relu"}, {"type": "dropout", "rate": 0.2}])
    net.fit(raw, NN_lim=192)
    ref = net.predict(raw)
    got = np.load(out)
    assert int(got["epochs"]) == net.trained_epochs
    assert np.array_equal(got["val"], np.array(net.history["val_loss"]))
    assert np.array_equal(got["imputed"], ref.values)


def test_bench_maps_one_synthetic_matrix_for_the_ranks_of_a_streamed_job(tmp_path):
    """bench.py --gpus N --stream (configs[4]): rank 0 generates the matrix into /dev/shm, every rank maps it -- the same function of
    (n, g, seed) as a private matrix, one inode."""
    code = (
        "import os, sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "r = int(os.environ['RANK'])\n"
        "rd = bench.FileRendezvous(r, 2)\n"
        "norm, seg = bench.shared_matrix(rd, 1300, 700, seed=0)\n"
        "ok = bool(np.array_equal(norm, bench.synth_counts(1300, 700, seed=0)))\n"
        "json.dump({'ok': ok, 'id': list(seg.identity), 'writable': bool(norm.flags.writeable)}, open(%r + str(r), 'w'))\n"
        "bench.file_vote(rd, 'bye', 0.0); rd.cleanup()\n"
    ) % (os.path.dirname(HERE), str(tmp_path / "out"))
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=300)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-2000:]
    a, b = (json.load(open(str(tmp_path / "out") + str(r))) for r in range(2))
    assert a["ok"] and b["ok"] and a["id"] == b["id"] and not b["writable"]
