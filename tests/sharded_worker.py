"""Worker of tests/test_sharded.py: one rank of a world_size-N gloo job (CPU).  The engine is
the CPU oracle injected into MultiNet's build() seam (helpers.multinet_with, test infrastructure); what is
under test is the sharding / early-stopping all-reduce / gather logic of deepimpute_amd.sharded
and MultiNet's comm path."""
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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    comm = None
    if world > 1:
        dist.init_process_group("gloo", init_method="env://")
        comm = TorchComm()
    rng = np.random.default_rng(3)
    n, g = 240, 360
    mu = rng.lognormal(0.5, 1.2, size=g)
    raw = pd.DataFrame(rng.poisson(rng.gamma(2.0, mu / 2.0, size=(n, g))).astype(np.float64),
                       index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
    default_dir = os.environ.get("SHARDED_WORKER_MODE") == "default_dir"      # no output_prefix: the per-process temporary default
    where = {} if default_dir else {"output_prefix": out_path + ".dir"}
    kw = dict(comm=comm, seed=17, sub_outputdim=64, ncores=1, verbose=0,
              architecture=[{"type": "dense", "neurons": 32, "activation": "relu"}, {"type": "dropout", "rate": 0.2}])
    net = multinet_with(OracleEngine, max_epochs=6, patience=2, learning_rate=2e-3, **where, **kw)
    net.fit(raw, NN_lim=300)
    imputed = net.predict(raw)
    rank = 0 if comm is None else comm.rank
    if default_dir:
        # a FRESH object of the same job finds the shards the fit wrote (every rank resolved the same shared directory)
        again = multinet_with(OracleEngine, **({"output_prefix": net.outputdir}), **kw)
        again.predictors, again.targets = net.predictors, net.targets
        reloaded = again.predict(raw)
        if rank == 0:
            assert np.array_equal(reloaded.values, imputed.values)
            assert sorted(f for f in os.listdir(net.outputdir) if f.startswith("model.rank")) == ["model.rank%d.npz" % r for r in range(world)]
    if rank == 0:
        np.savez(out_path, imputed=imputed.values, epochs=net.trained_epochs, val=np.array(net.history["val_loss"]),
                 loss=np.array(net.history["loss"]), K=len(net.predictors),
                 metrics=np.array([net.test_metrics["correlation"], net.test_metrics["MSE"]]))
    else:
        assert imputed is None and net.test_metrics is None
    if comm is not None:
        comm.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
