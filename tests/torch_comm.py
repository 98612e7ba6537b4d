"""gloo implementation of the deepimpute_amd.sharded Comm interface (TEST INFRASTRUCTURE: the CPU
suite runs world_size-2/3 jobs with it; the product's collectives are RCCL, deepimpute_amd.sharded.RcclComm)."""
import numpy as np


class TorchComm:
    """Collectives over an initialised torch.distributed process group (any backend; the
    test-suite uses gloo on CPU).  Arrays travel as host tensors."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self._group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def allreduce_sum(self, vec):
        import torch
        t = torch.tensor(np.asarray(vec, np.float64))
        self._dist.all_reduce(t, group=self._group)
        return t.numpy()

    def gather_predictions(self, engine, local_block, n_rows, counts, out_dim):
        import torch
        block = local_block() if callable(local_block) else local_block
        mine = torch.from_numpy(np.ascontiguousarray(block, np.float32))
        if self.rank == 0:
            parts = [torch.empty((n_rows, c * out_dim), dtype=torch.float32) for c in counts]
            self._dist.gather(mine, parts, dst=0, group=self._group)
            return np.hstack([p.numpy() for p in parts])
        self._dist.gather(mine, None, dst=0, group=self._group)
        return None

    def barrier(self):
        self._dist.barrier(group=self._group)

    def close(self):
        pass
