"""All ranks of a sharded job as THREADS of one process on one GPU (TEST INFRASTRUCTURE).  Implements the deepimpute_amd.sharded Comm
interface: the all-reduce is a rank-ordered sum behind a barrier, the gather is libdimn's dimn_comm_gather_loopback -- the root side of the
RCCL gather (arena sizing, block offsets, strided placement) with device-to-device copies where ncclRecv would run.  What it cannot stand
in for is RCCL itself."""
import threading

import numpy as np


class LoopbackWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.engines = [None] * world
        self.result = None


class LoopbackComm:
    device_gather = True        # predictions stay in HBM: predict_device + the (loop-back) gather

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def allreduce_sum(self, vec):
        s = self.shared
        s.slots[self.rank] = np.asarray(vec, np.float64).copy()
        s.barrier.wait()
        total = s.slots[0].copy()
        for r in range(1, self.world):              # rank order on every rank: the same bits everywhere
            total += s.slots[r]
        s.barrier.wait()                            # nobody overwrites a slot before everybody has summed
        return total

    def gather_predictions(self, engine, local_block, n_rows, counts, out_dim):
        from deepimpute_amd.engine import HipEngine
        s = self.shared
        s.engines[self.rank] = engine
        s.barrier.wait()                            # every rank's predict_device has been queued (the gather synchronises the streams)
        out = None
        if self.rank == 0:
            assert [e.K for e in s.engines] == list(counts)
            out = HipEngine.gather_loopback(s.engines, n_rows, root=0)
        s.barrier.wait()
        return out

    def barrier(self):
        self.shared.barrier.wait()

    def close(self):
        pass
