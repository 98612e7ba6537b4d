"""Sub-network sharding over the GPUs of one node (one process per GPU).

The K sub-networks of a MultiNet share no parameters (reference deepimpute/multinet.py:132-146
builds K disjoint branches); they are coupled only by the shared batch order and by ONE global
early-stopping decision on the summed validation loss (multinet.py:242-243).  So the path shards
by sub-network with no gradient traffic at all:

  * every rank plans identically (same seed -> same targets / predictors / split) and owns a
    contiguous block of sub-nets; Philox keys use the GLOBAL sub-net index, so the trained
    weights do not depend on the number of ranks;
  * per epoch: one all-reduce (sum) of two scalars -- validation and training loss -- so that
    every rank takes the same stop/continue decision the single-process reference would take;
  * at the end: the per-rank prediction blocks [cells, K_r*O] are gathered into the full
    [cells, K*O] matrix on rank 0 (RCCL send/recv straight to root over xGMI on the GPU path).

The collectives sit behind a tiny `Comm` interface (rank, world, allreduce_sum, gather_predictions,
barrier, close).  The product implementation is `RcclComm` (libdimn's dimn_comm_* on the GPU data
path; rendezvous through a private file under /tmp, no torch in the process); the CPU test-suite
plugs a gloo implementation of the same interface in (tests/torch_comm.py, world_size 2 and 3).
"""
import os
import time

import numpy as np


def shard_subnets(K, world, weights=None):
    """Contiguous split of K sub-nets over `world` ranks: (counts[r], offsets[r]).  Contiguous because Philox keys and the
    np.hstack order of the predictions use the global sub-net index.  Without weights: balanced by count.  With weights[k]
    (the predictor count D_k: a rank's step time and its X arena grow with its sum of D_k): the contiguous
    partition with the smallest maximum rank load (linear-partition dynamic programme); when the by-count split is within
    2 % of that optimum it is kept, so the common case of near-equal D_k shards exactly as before.  Ranks beyond K get zero
    sub-nets (callers must have K >= world for a useful job)."""
    by_count = [K // world + (1 if r < K % world else 0) for r in range(world)]
    counts = by_count
    if weights is not None and world > 1 and K > world:
        w = np.asarray(weights, np.float64)
        if w.shape != (K,) or not np.all(w > 0):
            raise ValueError("shard_subnets: weights must be K positive numbers")
        pre = np.concatenate([[0.0], np.cumsum(w)])
        load = lambda i, j: pre[j] - pre[i]                      # sub-nets [i, j)
        best = np.full((world + 1, K + 1), np.inf)
        cut = np.zeros((world + 1, K + 1), np.int64)
        best[0, 0] = 0.0
        for r in range(1, world + 1):
            for j in range(r, K - (world - r) + 1):             # every rank keeps at least one sub-net
                for i in range(r - 1, j):
                    cost = max(best[r - 1, i], load(i, j))
                    if cost < best[r, j]:                        # strict: ties keep the earliest cut (reproducible on every rank)
                        best[r, j], cut[r, j] = cost, i
        ends, j = [], K
        for r in range(world, 0, -1):
            ends.append(j)
            j = int(cut[r, j])
        ends = ends[::-1]
        optimal = [ends[0]] + [ends[r] - ends[r - 1] for r in range(1, world)]
        offs = np.concatenate([[0], np.cumsum(by_count)])
        worst_by_count = max(load(int(offs[r]), int(offs[r + 1])) for r in range(world))
        counts = by_count if worst_by_count <= 1.02 * best[world, K] else optimal
    offsets = [sum(counts[:r]) for r in range(world)]
    return counts, offsets


class SingleComm:
    rank, world = 0, 1

    def allreduce_sum(self, vec):
        return np.asarray(vec, np.float64)

    def gather_predictions(self, engine, local_block, n_rows, counts, out_dim):
        return local_block() if callable(local_block) else local_block

    def barrier(self):
        pass

    def close(self):
        pass


def _job_tag():
    """A name every rank of ONE job computes identically and no other job shares: the launcher's pid,
    its start time (field 22 of /proc/<pid>/stat: a recycled pid gets another one) and MASTER_PORT."""
    ppid = os.getppid()
    born = "0"
    try:
        with open("/proc/%d/stat" % ppid) as f:
            born = f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        pass
    return "%d_%s_%s" % (ppid, born, os.environ.get("MASTER_PORT", "0"))


class RcclComm:
    """RCCL over xGMI through libdimn (dimn_comm_*).  The 128-byte unique id is handed from rank 0 to
    the others through a file in a 0700 directory named after the job (`_job_tag`) plus a per-process
    sequence number, so that a second communicator of the same job (a second fit) never reads the
    first one's id; the GPU processes never import torch."""
    _seq = 0
    device_gather = True        # predictions stay in HBM: predict_device + dimn_comm_gather_predictions

    def __init__(self, engine, rank, world, tag=None, timeout=300.0):
        self.rank, self.world, self._engine = rank, world, engine
        RcclComm._seq += 1
        tag = tag or "%s_%d" % (_job_tag(), RcclComm._seq)
        self._dir = os.path.join("/tmp", "dimn_rdzv_%d_%s" % (os.getuid(), tag))
        os.makedirs(self._dir, mode=0o700, exist_ok=True)
        path = os.path.join(self._dir, "uid")
        if rank == 0:
            uid = engine.comm_unique_id()
            with open(path + ".tmp", "wb") as f:
                f.write(uid.tobytes())
            os.replace(path + ".tmp", path)
        else:
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > timeout:
                    raise TimeoutError("RCCL rendezvous file %s never appeared" % path)
                time.sleep(0.01)
            with open(path, "rb") as f:
                uid = np.frombuffer(f.read(), np.uint8)
        engine.comm_init(uid, world, rank)
        self._open = True

    def allreduce_sum(self, vec):
        return self._engine.comm_allreduce_sum(vec)

    def gather_predictions(self, engine, local_block, n_rows, counts, out_dim):
        # the local block is already in HBM (predict_device); root receives into HBM and copies out
        return engine.comm_gather_predictions(n_rows, counts, root=0, is_root=self.rank == 0)

    def barrier(self):
        self._engine.comm_allreduce_sum(np.zeros(1))

    def close(self):
        """Collective: every rank has passed the barrier (hence read the id) before root removes it."""
        if not self._open:
            return
        self._open = False
        self.barrier()
        self._engine.comm_destroy()
        if self.rank == 0:
            import shutil
            shutil.rmtree(self._dir, ignore_errors=True)


def fit_sharded(engine, comm, max_epochs, patience):
    """model.fit with EarlyStopping(monitor='val_loss', patience) over a sharded job: each rank
    trains its sub-nets; the monitored quantity is the sum over ALL sub-nets (one scalar
    all-reduce per epoch), so every rank stops at the same epoch the single-process run would
    (reference multinet.py:238-246; Keras: strict <, min_delta 0, last-epoch weights)."""
    best, wait = np.inf, 0
    loss_hist, val_hist = [], []
    epoch = 0
    while epoch < max_epochs:
        tl = engine.train_epoch(epoch)
        vl = engine.val_loss()
        tot = comm.allreduce_sum(np.array([np.sum(tl), np.sum(vl)], np.float64))
        loss_hist.append(float(tot[0]))
        val_hist.append(float(tot[1]))
        epoch += 1
        if tot[1] < best:
            best, wait = tot[1], 0
        else:
            wait += 1
            if wait >= patience:
                break
    return epoch, np.array(loss_hist), np.array(val_hist)


def predict_sharded(engine, comm, counts, rows=None, n_rows=None):
    """model.predict over a sharded job: rank 0 gets np.hstack over ALL sub-nets, others None."""
    if getattr(comm, "device_gather", False):
        engine.predict_device(rows, n_rows)
        n = len(rows) if rows is not None else (n_rows if n_rows is not None else engine.n_cells)
        return comm.gather_predictions(engine, None, n, counts, engine.O)
    block = engine.predict(rows, n_rows)
    return comm.gather_predictions(engine, block, block.shape[0], counts, engine.O)
