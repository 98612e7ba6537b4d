"""Worker of tests/test_shm.py: one rank of a world_size-2 gloo job (CPU) whose count frame exists ONCE on the node
(deepimpute_amd._shm.share_frame): rank 0 makes the frame, rank 1 passes None; both fit and predict through MultiNet's comm path
on the CPU oracle (test infrastructure, helpers.multinet_with) and report what the kernel says about their mappings."""
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_path):
    import torch.distributed as dist
    from helpers import multinet_with
    from torch_comm import TorchComm
    from oracle.dimo import OracleEngine
    from deepimpute_amd import _shm

    dist.init_process_group("gloo", init_method="env://")
    comm = TorchComm()
    n, g = 4096, 2048                                         # 64 MB as float64: large against a page, small against the suite's budget
    raw = None
    if comm.rank == 0:
        rng = np.random.default_rng(3)
        mu = rng.lognormal(0.5, 1.2, size=g)
        raw = pd.DataFrame(rng.poisson(rng.gamma(2.0, mu / 2.0, size=(n, g))).astype(np.float64),
                           index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
    node = _shm.NodeComm()                                                   # (what a product job uses before its engine exists: RANK / WORLD_SIZE + file barrier)
    assert (node.rank, node.world) == (comm.rank, comm.world)
    shared = _shm.share_frame(raw, node)
    node.close()
    seg = _shm.shared_of(shared.values)
    assert seg is not None and not os.path.exists(seg.path)                  # the name is gone as soon as every rank has mapped it
    if comm.rank == 0:
        assert np.array_equal(shared.values, raw.values) and shared.index.equals(raw.index) and shared.columns.equals(raw.columns)
        del raw                                                               # the private copy can go: the job reads the segment
    else:
        assert not shared.values.flags.writeable
    _ = float(shared.values.sum())                                           # touch every page
    report = {"rank": comm.rank, "identity": list(seg.identity), "frame_smaps": seg.smaps(), "frame_bytes": int(shared.values.nbytes)}
    kw = dict(comm=comm, seed=17, sub_outputdim=64, ncores=1, verbose=0, output_prefix=out_path + ".dir",
              architecture=[{"type": "dense", "neurons": 32, "activation": "relu"}, {"type": "dropout", "rate": 0.2}])
    net = multinet_with(OracleEngine, max_epochs=2, patience=2, learning_rate=2e-3, **kw)
    captured = {}
    real = _shm.shared_log1p

    def spy(frame, c):                                                        # fit() and predict() must take the shared log1p matrix, not a private one
        out = real(frame, c)
        seg_n = _shm.shared_of(out)
        assert seg_n is not None
        _ = float(out.sum())                                                  # touch every page
        captured.setdefault("norm", []).append({"identity": list(seg_n.identity), "smaps": seg_n.smaps(),
                                                "equal": bool(np.array_equal(out, np.log1p(frame.values).astype(np.float32))), "seg": seg_n})
        return out
    _shm.shared_log1p = spy
    net.fit(shared, NN_lim=192)
    imputed = net.predict(shared)
    assert len(captured["norm"]) == 2
    assert all(c["seg"].array is None for c in captured["norm"])             # fit() / predict() released their segment (tmpfs pages live until every rank unmaps)
    report["norm_identity"] = captured["norm"][0]["identity"]
    report["norm_smaps"] = captured["norm"][0]["smaps"]
    report["norm_equal"] = captured["norm"][0]["equal"] and captured["norm"][1]["equal"]
    if comm.rank == 0:
        np.savez(out_path, imputed=imputed.values, epochs=net.trained_epochs, val=np.array(net.history["val_loss"]))
    else:
        assert imputed is None
    with open("%s.rank%d.json" % (out_path, comm.rank), "w") as f:
        json.dump(report, f)
    comm.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
