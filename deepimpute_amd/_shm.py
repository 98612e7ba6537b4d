"""ONE physical host copy of a [cells, genes] matrix for all ranks of a one-node job (one process per GPU).

The reference materialises the frame -- and `norm_data = np.log1p(raw).astype(np.float32)`, multinet.py:217, and the 4K
gathered copies of :231-235 -- inside its single process.  A sharded job has one process per GPU, and "every rank plans
identically" (sharded.py) would otherwise mean every rank HOLDS the frame: BASELINE configs[4] (1M cells x 30k genes) is
240 GB as a float64 frame + 120 GB as the float32 log1p matrix, per rank -- 2.9 TB at eight ranks, and eight streamed
hand-overs reading eight different copies.  Here the ranks of a node share them through POSIX shared memory:

  * `share_frame(raw_or_None, comm)`: rank 0 passes its DataFrame, every other rank None; all of them get back a DataFrame
    over ONE /dev/shm segment (labels travel through a small side file; other ranks map the values read-only).  The name is
    unlinked as soon as every rank has mapped it: nothing is left behind whatever happens to the job afterwards.
  * `shared_log1p(raw, comm)`: float32(log1p(raw)) into a second segment, each rank filling its own slice of the rows;
    MultiNet.fit() calls it for a sharded job whose frame came from `share_frame` (or when DIMN_SHARE_NORM=1), so the
    streamed hand-over of every rank (dimn_set_matrix_streamed, rotated by dimn_set_stream_order) reads the same pages.

Host memory of the 8-rank configs[4] job: 240 + 120 GB once instead of eight times.  The only collective needed is barrier():
sharded's Comm objects have it, and `NodeComm` provides it before an engine exists (the product's RCCL communicator is bound to the
engine inside fit())."""
import mmap
import os
import pickle

import numpy as np
import pandas as pd

_SEQ = [0]


def _segment_name(tag):
    from .sharded import _job_tag
    _SEQ[0] += 1
    return "dimn_%d_%s_%s_%d" % (os.getuid(), _job_tag(), tag, _SEQ[0])


def _shm_dir():
    return os.environ.get("DIMN_SHM_DIR", "/dev/shm")


class NodeComm:
    """rank / world / barrier() for the ranks of ONE node before any engine (hence any RCCL communicator) exists -- what share_frame needs
    when the job's communicator is the product's (`MultiNet(comm="rccl")` binds RCCL to the engine inside fit()).  Rank and world size
    come from the launcher's environment (RANK / WORLD_SIZE: torchrun); a barrier is one file per rank in a 0700 directory named after
    the job.  Not a data path."""

    def __init__(self, rank=None, world=None, timeout=600.0):
        import time
        from .sharded import _job_tag
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self._dir = os.path.join("/tmp", "dimn_node_%d_%s" % (os.getuid(), _job_tag()))
        os.makedirs(self._dir, mode=0o700, exist_ok=True)
        self._n, self._timeout, self._time = 0, timeout, time
        self._closed = False
        if self.rank == 0:                                       # (a job that never calls close(): the directory still goes when rank 0 exits)
            import atexit
            atexit.register(self._sweep)

    def _sweep(self):
        if not self._closed:
            import shutil
            shutil.rmtree(self._dir, ignore_errors=True)

    def barrier(self):
        self._n += 1
        open(os.path.join(self._dir, "b%d_%d" % (self._n, self.rank)), "w").close()
        t0 = self._time.time()
        for r in range(self.world):
            q = os.path.join(self._dir, "b%d_%d" % (self._n, r))
            while not os.path.exists(q):
                if self._time.time() - t0 > self._timeout:
                    raise TimeoutError("NodeComm.barrier: rank %d never arrived" % r)
                self._time.sleep(0.001)

    def close(self, timeout=20.0):
        """Collective.  Rank 0 removes the directory only after every other rank has left the last barrier (a rank still polling for
        a file of that barrier must not find the directory gone), or `timeout` seconds later."""
        self.barrier()
        if self.rank != 0:
            try:
                open(os.path.join(self._dir, "left_%d" % self.rank), "w").close()
            except OSError:
                pass
        else:
            t0 = self._time.time()
            while self._time.time() - t0 < timeout and not all(os.path.exists(os.path.join(self._dir, "left_%d" % r)) for r in range(1, self.world)):
                self._time.sleep(0.002)
            self._sweep()
        self._closed = True


class SharedArray:
    """A [rows, cols] array in a /dev/shm segment mapped by every rank of the job.  `array` is a numpy view of the mapping
    (read-only on ranks that only read); `identity` = (st_dev, st_ino) of the segment, the same on every rank."""

    def __init__(self, comm, shape, dtype, tag, writers="root"):
        self.shape, self.dtype = tuple(int(x) for x in shape), np.dtype(dtype)
        nbytes = max(1, int(np.prod(self.shape)) * self.dtype.itemsize)
        rank = comm.rank
        # every rank computes the same name (job tag + a per-process sequence number that advances identically on all of them)
        path = os.path.join(_shm_dir(), _segment_name(tag))
        self.path = path
        if rank == 0:
            fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_EXCL, 0o600)
            try:
                os.ftruncate(fd, nbytes)
            except BaseException:
                os.close(fd)
                os.unlink(path)
                raise
        comm.barrier()                                   # the segment exists and has its size
        writable = rank == 0 or writers == "all"
        if rank != 0:
            fd = os.open(path, os.O_RDWR if writable else os.O_RDONLY)
        try:
            info = os.fstat(fd)
            self.identity = (info.st_dev, info.st_ino)
            self._map = mmap.mmap(fd, nbytes, mmap.MAP_SHARED, mmap.PROT_READ | (mmap.PROT_WRITE if writable else 0))
        finally:
            os.close(fd)
        comm.barrier()                                   # every rank has mapped it ...
        if rank == 0:
            os.unlink(path)                              # ... so the name can go: the pages live as long as a mapping does
        self.array = np.frombuffer(self._map, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)
        self._comm = comm

    def publish(self):
        """Collective: what the writers wrote is visible to every rank (same pages: a barrier is all it takes)."""
        self._comm.barrier()

    def smaps(self):
        """The kernel's accounting of THIS process' mapping (/proc/self/smaps): {'Size', 'Rss', 'Pss', 'Private_Dirty', ...} in kB."""
        lo = np.frombuffer(self._map, np.uint8, 1).ctypes.data if self._map.size() else 0   # (works for read-only maps too)
        out, inside = {}, False
        with open("/proc/self/smaps") as f:
            for line in f:
                head = line.split()
                if "-" in head[0] and len(head) >= 5 and ":" not in head[0]:
                    a, b = (int(x, 16) for x in head[0].split("-"))
                    inside = a <= lo < b
                elif inside and line.rstrip().endswith("kB"):
                    out[head[0].rstrip(":")] = int(head[1])
        return out


def _copy_rows(dst, src, lo, hi, log1p=False):
    from . import _hostpar
    step = max(1, (64 << 20) // max(1, src.shape[1] * 8))

    def work(r0):
        r1 = min(hi, r0 + step)
        block = src[r0:r1]
        dst[r0:r1] = np.log1p(block) if log1p else block  # (assignment casts to dst's dtype: float32(log1p(float64)), as multinet.py:217)
    starts = list(range(lo, hi, step))
    if starts:
        _hostpar.pmap(work, starts)


_BY_ADDRESS = {}              # first byte of a shared mapping -> its SharedArray (pandas / numpy drop view attributes; the address survives them)


def _wrap(shared):
    _BY_ADDRESS[shared.array.__array_interface__["data"][0]] = shared
    return shared.array


def shared_of(values):
    """The SharedArray behind `values` (a frame's .values made by share_frame / shared_log1p), or None."""
    try:
        return _BY_ADDRESS.get(values.__array_interface__["data"][0])
    except (AttributeError, TypeError):
        return None


def release(values):
    """Forget the segment behind `values`; it is unmapped once the last numpy view of it is gone (tmpfs pages live until EVERY rank has
    unmapped them: a long-lived process should release what it no longer reads -- fit() / predict() do so for the log1p matrix)."""
    shared = shared_of(values)
    if shared is not None:
        _BY_ADDRESS.pop(shared.array.__array_interface__["data"][0], None)
        shared.array = None


def share_frame(raw, comm, dtype=None):
    """Collective over the ranks of a one-node job.  Rank 0 passes the count frame, every other rank None; every rank gets a
    DataFrame with the same labels whose values are ONE /dev/shm segment (read-only on ranks > 0).  dtype: the segment's element
    type (default: the frame's own; np.float32 halves the segment and is exact for counts below 2**24)."""
    rank = comm.rank
    if comm.world == 1:
        return raw
    if (raw is None) == (rank == 0):
        raise ValueError("share_frame: rank 0 passes the frame, every other rank None")
    meta_path = os.path.join(_shm_dir(), _segment_name("labels"))
    if rank == 0:
        values = raw.values
        if values.dtype.kind not in "fiu" or values.ndim != 2:
            raise TypeError("share_frame: a numeric [cells, genes] frame is expected (got dtype %s)" % values.dtype)
        meta = {"shape": values.shape, "dtype": np.dtype(dtype or values.dtype).str, "index": raw.index, "columns": raw.columns}
        fd = os.open(meta_path, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            pickle.dump(meta, f, protocol=pickle.HIGHEST_PROTOCOL)
    comm.barrier()
    if rank != 0:
        with open(meta_path, "rb") as f:                 # (a 0600 file of this user, written by rank 0 of this job a moment ago)
            meta = pickle.load(f)
    comm.barrier()
    if rank == 0:
        os.unlink(meta_path)
    shared = SharedArray(comm, meta["shape"], np.dtype(meta["dtype"]), "frame")
    if rank == 0:
        _copy_rows(shared.array, raw.values, 0, meta["shape"][0])
    shared.publish()
    return pd.DataFrame(_wrap(shared), index=meta["index"], columns=meta["columns"], copy=False)


def shared_log1p(raw, comm):
    """Collective: float32(log1p(raw.values)) (multinet.py:217) in ONE segment; rank r computes rows [r n / w, (r + 1) n / w).
    Every rank must hold the same `raw` (share_frame's, or its own equal copy).  Returns the [cells, genes] float32 array."""
    n, g = raw.shape
    shared = SharedArray(comm, (n, g), np.float32, "norm", writers="all")
    lo, hi = n * comm.rank // comm.world, n * (comm.rank + 1) // comm.world
    _copy_rows(shared.array, raw.values, lo, hi, log1p=True)
    shared.publish()
    return _wrap(shared)
