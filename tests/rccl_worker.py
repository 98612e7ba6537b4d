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


def main(out_path):
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
