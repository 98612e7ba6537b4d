"""Worker of tests/test_gpu_rccl_multi.py: ONE rank of a one-process-per-GPU job on the HIP engine with RCCL
(MultiNet(comm="rccl") and bench.impute_once).  Launched with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT set;
no torch in this process.  Rank 0 writes the results to the .npz given as argv[1]."""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def big_shapes(out_path):
    """BASELINE configs[3] shapes on the wire: the K = 40 sub-nets of the 50k x 20k job (D_k ~ 2 400) sharded over the ranks --
    20 per rank at world 2 -- two optimiser steps, validation, predict of 1 024 cells and the strided placement of every
    rank's [n][K_r * 512] block into root's [n][40 * 512] matrix."""
    import bench
    from deepimpute_amd.engine import HipEngine
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cfg = dict(bench.CONFIGS["cfg3"], n=1024)
    norm = bench.synth_counts(cfg["n"], cfg["g"], seed=0, threads=8)
    targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=0)
    K = targets.shape[0]
    counts, offs = bench.shard(K, world)
    train, val = np.arange(0, 64 + 37, dtype=np.int32), np.arange(900, 1024, dtype=np.int32)
    eng = bench.make_engine(HipEngine, cfg, targets, preds, norm, train, val, counts, offs, rank, local, 1e-3)
    rdzv = bench.FileRendezvous(rank, world)
    comm, err = bench.bring_up_rccl(eng, rdzv, rank, world)
    assert comm is not None, err
    assert eng.comm_info() == (world, rank)
    vsum = bench.impute_once(eng, 1, comm, counts, cfg["n"])
    full = eng.comm_gather_predictions(cfg["n"], counts, root=0, is_root=rank == 0)
    stats = (comm.gather_bytes, comm.gathers)
    comm.allreduce_sum(np.zeros(1))
    comm.close()
    rdzv.cleanup()
    eng.close()
    if rank == 0:
        assert full.shape == (cfg["n"], K * cfg["O"])
        assert stats == (4 * cfg["n"] * cfg["O"] * (K - counts[0]), 1)
        np.savez(out_path, vsum=vsum, full=full[::3])
    else:
        assert full is None


def main(out_path):
    if os.environ.get("DIMN_RCCL_WORKER_MODE") == "big":
        return big_shapes(out_path)
    import bench
    from deepimpute_amd.engine import HipEngine
    from deepimpute_amd.multinet import MultiNet
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    # --- the drop-in surface: MultiNet(comm="rccl") -------------------------------------------------
    rng = np.random.default_rng(3)
    n, g = 240, 640
    mu = rng.lognormal(0.5, 1.2, size=g)
    raw = pd.DataFrame(rng.poisson(rng.gamma(2.0, mu / 2.0, size=(n, g))).astype(np.float64),
                       index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
    net = MultiNet(comm="rccl", seed=17, sub_outputdim=64, ncores=1, verbose=0, max_epochs=6, patience=2,
                   learning_rate=2e-3, output_prefix=out_path + ".dir",
                   architecture=[{"type": "dense", "neurons": 32, "activation": "relu"}, {"type": "dropout", "rate": 0.2}])
    net.fit(raw, NN_lim=512)                       # K = 9 sub-nets (filter_genes adds one): uneven shards at world 2, 4, 8
    assert net.device_id == local
    imputed = net.predict(raw)
    # a fresh object reloads the per-rank shards written by fit()
    again = MultiNet(comm="rccl", seed=17, sub_outputdim=64, ncores=1, verbose=0, output_prefix=out_path + ".dir")
    again.predictors, again.targets = net.predictors, net.targets
    reloaded = again.predict(raw)
    again.close()
    net.close()

    # --- the bench's timed unit over RCCL (what `bench.py --gpus N` runs) ---------------------------
    cfg = dict(n=700, g=4200, H=256, O=512, B=64)        # K = 9 sub-nets of D ~ 1850: every rank of an 8-GPU job owns one
    norm = bench.synth_counts(cfg["n"], cfg["g"], seed=0, threads=4)
    targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=0)
    train, val = bench.split_rows(cfg["n"], seed=0)
    K = targets.shape[0]
    counts, offs = bench.shard(K, world)
    eng = bench.make_engine(HipEngine, cfg, targets, preds, norm, train[:64 * 6 + 5], val, counts, offs, rank, local, 1e-4)
    rdzv = bench.FileRendezvous(rank, world)
    uid = rdzv.broadcast_bytes("uid", eng.comm_unique_id().tobytes() if rank == 0 else None)
    eng.comm_init(np.frombuffer(uid, np.uint8), world, rank)
    vsum = bench.impute_once(eng, 2, bench.RcclBenchComm(eng), counts, cfg["n"])
    eng.predict_device()
    full = eng.comm_gather_predictions(cfg["n"], counts, root=0, is_root=rank == 0)
    eng.comm_allreduce_sum(np.zeros(1))
    eng.comm_destroy()
    rdzv.cleanup()
    eng.close()

    if rank == 0:
        assert imputed is not None and reloaded is not None and full is not None
        np.savez(out_path, imputed=imputed.values, reloaded=reloaded.values, epochs=net.trained_epochs,
                 val=np.array(net.history["val_loss"]), loss=np.array(net.history["loss"]), K=len(net.predictors),
                 metrics=np.array([net.test_metrics["correlation"], net.test_metrics["MSE"]]),
                 vsum=vsum, full=full[::37])
    else:
        assert imputed is None and reloaded is None and net.test_metrics is None and full is None


if __name__ == "__main__":
    main(sys.argv[1])
