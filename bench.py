#!/usr/bin/env python
"""bench.py -- cells/sec of the MultiNet hot path (fit + predict) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1: launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per
GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* are read from the environment).

A *step* is one end-to-end impute of the synthetic matrix through the hot path with the inputs
(log1p matrix + predictor/target index lists) already resident: device gather of the K
(cells x predictors)/(cells x targets) blocks, weight init, E epochs of training (each followed
by the validation pass, as model.fit does) and the forward pass over all cells, results left in
HBM (N>1: gathered into root's HBM over RCCL).  value = n_cells / seconds per step.

Workload = BASELINE.json configs[2] (the config the metric is quoted on; it fits one GPU):
50k cells x 20k genes, K=40 sub-nets, H=256, O=512, batch 64, dropout 0.2, Adam lr 1e-4.
Early stopping makes the epoch count data dependent, so the bench fixes E (--epochs) on every
leg; DESIGN.md records what the early-stopped run takes on this generator.

No torch in this process: the GPU work goes through libdimn.so (ctypes); multi-rank control
uses a file rendezvous under /tmp and RCCL itself (libdimn's dimn_comm_*).
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (cells, genes, hidden, out, batch)
    "cfg3": dict(n=50000, g=20000, H=256, O=512, B=64, label="50k cells x 20k genes, K=40 sub-nets, H=256, O=512, batch 64"),
    # BASELINE configs[4]: one rank's share with --limit-subnets 8 (K = 59 sub-nets over 8 GPUs), --precision bf16 --stream
    "cfg5": dict(n=1000000, g=30000, H=256, O=512, B=64, label="1M cells x 30k genes (streamed from host), K=59 sub-nets, H=256, O=512, batch 64"),
    "cfg2": dict(n=5000, g=5000, H=256, O=512, B=64, label="5k cells x 5k genes, K=10 sub-nets, H=256, O=512, batch 64"),
    "tiny": dict(n=2000, g=1500, H=256, O=512, B=64, label="2k cells x 1.5k genes (plumbing check)"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3


def synth_counts(n, g, seed=0, threads=None, out=None):
    """BASELINE.md generator: mu_j ~ LogNormal(0.5,1.2); lambda_ij ~ Gamma(2, mu_j/2);
    X_ij ~ Poisson(lambda_ij).  Row blocks are drawn by independent child streams in threads
    (numpy releases the GIL), so the matrix is a pure function of (n, g, seed)."""
    threads = threads or min(32, os.cpu_count() or 1)
    root = np.random.default_rng(seed)
    mu = root.lognormal(0.5, 1.2, size=g)
    out = np.empty((n, g), np.float32) if out is None else out       # (out: a caller's [n, g] float32 array, e.g. a shared segment)
    blk = 512
    starts = list(range(0, n, blk))
    seeds = np.random.SeedSequence(seed).spawn(len(starts))

    def fill(i):
        r = np.random.default_rng(seeds[i])
        a, b = starts[i], min(n, starts[i] + blk)
        lam = r.gamma(2.0, mu / 2.0, size=(b - a, g))
        out[a:b] = np.log1p(r.poisson(lam)).astype(np.float32)    # log1p matrix (multinet.py:217)

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(fill, range(len(starts))))
    return out


def synth_indices(g, O, seed=0, ntop=5):
    """Target partition as filter_genes/setTargets produce with NN_lim=g (K = ceil(g/O), the
    last slots filled with random genes, multinet.py:312-342) and predictor lists of the size
    setPredictors yields (union of ntop picks per target over non-target genes: D_k ~
    (g-O)(1-exp(-ntop*O/(g-O))), BASELINE.md section 2 probe: 2373..2418 at g=20k)."""
    rng = np.random.default_rng(seed + 1)
    K = -(-g // O)
    genes = rng.permutation(g)
    fill = rng.integers(0, g, size=K * O - g)
    targets = np.concatenate([genes, fill]).reshape(K, O).astype(np.int32)
    preds = []
    for k in range(K):
        non = np.setdiff1d(np.arange(g, dtype=np.int32), targets[k])
        picks = rng.integers(0, non.size, size=ntop * O)
        _, first = np.unique(picks, return_index=True)
        preds.append(non[picks[np.sort(first)]].astype(np.int32))
    return targets, preds


def split_rows(n, seed=0):
    rng = np.random.default_rng(seed + 2)
    val = rng.choice(n, int(0.05 * n), replace=False).astype(np.int32)     # multinet.py:228
    train = np.setdiff1d(np.arange(n, dtype=np.int32), val).astype(np.int32)
    return train, val


def shard(K, world):
    """Contiguous, balanced split of sub-nets over ranks: counts[r], offsets[r]."""
    counts = [K // world + (1 if r < K % world else 0) for r in range(world)]
    offs = [sum(counts[:r]) for r in range(world)]
    return counts, offs


class FileRendezvous:
    """Single-node rendezvous for the RCCL unique id (all ranks are children of one torchrun
    agent -> the parent pid + MASTER_PORT name a directory nobody else uses)."""

    def __init__(self, rank, world):
        from deepimpute_amd.sharded import _job_tag
        # launcher pid + its start time + MASTER_PORT: a name no earlier job can have left a stale id under
        self.dir = os.path.join("/tmp", "dimn_rdzv_%d_bench_%s" % (os.getuid(), _job_tag()))
        self.rank, self.world = rank, world
        os.makedirs(self.dir, mode=0o700, exist_ok=True)

    def broadcast_bytes(self, name, payload=None, timeout=300.0):
        path = os.path.join(self.dir, name)
        if self.rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(payload)
            os.replace(tmp, path)
            return payload
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout:
                raise TimeoutError("rendezvous: %s never appeared" % path)
            time.sleep(0.01)
        with open(path, "rb") as f:
            return f.read()

    def cleanup(self, timeout=20.0):
        """Every rank calls this when it no longer reads the directory; rank 0 removes it once the others have said so (or `timeout`
        seconds later: a rank that died must not keep it forever).  Without the wait a rank still polling for a file of the last vote
        could find the directory gone and spin until its own timeout."""
        if self.rank != 0:
            try:
                open(os.path.join(self.dir, "left_%d" % self.rank), "w").close()
            except OSError:
                pass                                             # (already removed: rank 0 gave up waiting)
            return
        t0 = time.time()
        while time.time() - t0 < timeout and not all(os.path.exists(os.path.join(self.dir, "left_%d" % r)) for r in range(1, self.world)):
            time.sleep(0.002)
        import shutil
        shutil.rmtree(self.dir, ignore_errors=True)


class RcclBenchComm:
    """The job's collectives on RCCL over xGMI (libdimn's dimn_comm_*), with host clocks around them (both calls block until
    the collective has finished on the handle's stream)."""
    kind = "rccl"

    def __init__(self, eng):
        self.eng = eng
        self.reset()

    def reset(self):
        self.allreduce_s, self.allreduces, self.gather_s, self.gathers, self.gather_bytes = 0.0, 0, 0.0, 0, 0

    def allreduce_sum(self, vec, timed=True):
        t0 = time.perf_counter()
        out = self.eng.comm_allreduce_sum(vec)
        if timed:
            self.allreduce_s += time.perf_counter() - t0
            self.allreduces += 1
        return out

    def gather(self, n, counts):
        t0 = time.perf_counter()
        self.eng.comm_gather_predictions(n, counts, root=0, is_root=False)       # result stays in root's HBM
        self.gather_s += time.perf_counter() - t0
        self.gathers += 1
        # bytes that cross xGMI into root: every peer's [n][K_r * O] fp32 block
        self.gather_bytes += 4 * n * self.eng.O * (int(np.sum(counts)) - int(counts[0]))

    def close(self):
        self.eng.comm_destroy()


def file_vote(rdzv, name, value, timeout=600.0):
    """Sum of one number per rank through files of the rendezvous directory.  Used ONLY to agree on whether RCCL came up on
    every rank (a rank whose ncclCommInitRank failed cannot take part in an RCCL collective to say so); never a data path."""
    path = os.path.join(rdzv.dir, "%s_%d" % (name, rdzv.rank))
    with open(path + ".tmp", "w") as f:
        f.write(repr(float(value)))
    os.replace(path + ".tmp", path)
    total = 0.0
    for r in range(rdzv.world):
        q = os.path.join(rdzv.dir, "%s_%d" % (name, r))
        t0 = time.time()
        while not os.path.exists(q):
            if time.time() - t0 > timeout:
                raise TimeoutError("rendezvous vote %s: rank %d never arrived" % (name, r))
            time.sleep(0.001)
        with open(q) as f:
            total += float(f.read())
    return total


class VoteComm:
    """barrier() for deepimpute_amd._shm before RCCL is up (the matrix is made before the engines exist): one file vote per barrier."""

    def __init__(self, rdzv):
        self.rdzv, self.rank, self.world, self._n = rdzv, rdzv.rank, rdzv.world, 0

    def barrier(self):
        self._n += 1
        file_vote(self.rdzv, "shm%d" % self._n, 0.0)


def shared_matrix(rdzv, n, g, seed=0):
    """The synthetic log1p matrix ONCE per node (a streamed N-rank job: configs[4] is 120 GB -- eight private copies would be 1 TB of host
    memory and eight times the generator's work): rank 0 generates it into a /dev/shm segment, every rank maps it."""
    from deepimpute_amd import _shm
    comm = VoteComm(rdzv)
    seg = _shm.SharedArray(comm, (n, g), np.float32, "benchnorm")
    if rdzv.rank == 0:
        synth_counts(n, g, seed=seed, out=seg.array)             # the same function of (n, g, seed) as a private matrix
    seg.publish()
    return seg.array, seg


def bring_up_rccl(eng, rdzv, rank, world):
    """(RcclBenchComm, None) when every rank holds a working RCCL communicator, else (None, error string) on EVERY rank.
    There is no other transport: without RCCL an N > 1 job has no gather and no global early-stopping quantity, so the
    bench reports value null (rccl_failure_line) instead of a number."""
    err = None
    # RCCL prints a version banner on stdout at init; stdout must carry exactly one JSON line
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        uid = rdzv.broadcast_bytes("uid", eng.comm_unique_id().tobytes() if rank == 0 else None)
        eng.comm_init(np.frombuffer(uid, np.uint8), world, rank)
        eng.comm_allreduce_sum(np.zeros(1))
        nranks, myrank = eng.comm_info()
        if (nranks, myrank) != (world, rank):
            err = "communicator reports ranks=%d rank=%d, launched as %d / %d" % (nranks, myrank, world, rank)
    except Exception as e:
        err = repr(e)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    failed = file_vote(rdzv, "rccl_vote", 0.0 if err is None else 1.0)
    if failed == 0:
        return RcclBenchComm(eng), None
    return None, err or "RCCL failed on %d other rank(s)" % int(failed)


def bring_up_job(factory, rdzv, rank, world):
    """(engine, RcclBenchComm, None), or (engine or None, None, error string) on EVERY rank.  Anything that can go wrong before the
    first collective belongs here: a rank whose engine cannot be built (dimn_create on a device that does not exist: --gpus 2 on a
    one-GPU box) must not leave the others waiting inside ncclCommInitRank, so the ranks first agree through files that every
    engine exists, and only then bring RCCL up (bring_up_rccl, which votes again)."""
    eng, err = None, None
    try:
        eng = factory()
    except Exception as e:
        err = "engine: %r" % (e,)
    failed = file_vote(rdzv, "engine_vote", 0.0 if err is None else 1.0)
    if failed:
        return eng, None, err or "the engine could not be built on %d other rank(s)" % int(failed)
    comm, rccl_error = bring_up_rccl(eng, rdzv, rank, world)
    return eng, comm, rccl_error


def rccl_failure_line(args, world, label, error):
    """The line of an N > 1 run whose RCCL communicator did not come up: the contract's keys with value null, so that nothing
    downstream can mistake it for a measurement."""
    return {"metric": "cells/sec end-to-end impute (fit+predict)", "value": None, "unit": "cells/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": label, "parallelism": "subnets sharded x%d" % world, "collectives": None},
            "rccl_error": error, "roofline": None, "cpu_baseline": None}


def make_engine(cls, cfg, targets, preds, norm, train, val, counts, offs, rank, device_id, lr, stream=False, **kw):
    ks = range(offs[rank], offs[rank] + counts[rank])
    eng = cls([len(preds[k]) for k in ks], cfg["H"], cfg["O"], batch_size=cfg["B"], dropout_rate=0.2,
              learning_rate=lr, seed=1234, device_id=device_id, subnet_offset=offs[rank], **kw)
    for i, k in enumerate(ks):
        eng.set_indices(i, preds[k], targets[k])
    if stream:
        eng._bench_norm = norm                       # streamed hand-over: the matrix stays on the host, every impute re-streams it
        if len(counts) > 1 and hasattr(eng, "set_stream_order"):
            eng.set_stream_order(rank, len(counts))  # the ranks read one shared copy: each starts at another row block
    else:
        eng.set_matrix(norm)
    eng.n_cells = norm.shape[0]
    eng.set_split_later = (train, val)
    if not stream:
        eng.set_split(train, val)
    return eng


def impute_once(eng, epochs, comm=None, counts=None, n=None):
    """The timed unit: gather -> init -> E x (train epoch + validation) -> predict (-> gather).  With a streamed matrix
    (--stream) the hand-over IS the gather: the matrix crosses PCIe in row blocks inside the timed region."""
    if getattr(eng, "_bench_norm", None) is not None:
        t_h = time.perf_counter()
        eng.set_matrix(eng._bench_norm, streamed=True, with_targets=True)
        eng.synchronize()
        eng._bench_handover_s = time.perf_counter() - t_h       # the matrix crosses PCIe in row blocks, each gathered on the device as it lands
        eng.set_split(*eng.set_split_later)
    else:
        eng.gather(True)
    eng.init_weights()
    vsum = 0.0
    ident = np.arange(eng.n_train, dtype=np.int32) if os.environ.get("DIMN_BENCH_IDENTITY_PERM") else None   # diagnostic only
    for e in range(epochs):
        t_e = time.perf_counter()
        eng.train_epoch(e, ident)
        eng._bench_train_s = getattr(eng, "_bench_train_s", 0.0) + time.perf_counter() - t_e      # (train_epoch returns the losses: it has synchronised)
        eng._bench_train_steps = getattr(eng, "_bench_train_steps", 0) + -(-eng.n_train // eng.B)
        v = eng.val_loss()
        if comm is not None:                       # global early-stopping quantity (multinet.py:242-243)
            v = comm.allreduce_sum(np.array([v.sum()]))
        vsum = float(np.sum(v))
    t_p = time.perf_counter()
    eng.predict_device()
    eng.synchronize()
    eng._bench_predict_s = time.perf_counter() - t_p
    if comm is not None:
        comm.gather(n, counts)
    eng.synchronize()
    return vsum


def measured_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary
    (profiles/*_traffic.json, made by tools/pmc_traffic.py from separate rocprofv3 --pmc passes);
    None when no summary for this kernel is committed."""
    import glob
    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json"))):
        try:
            with open(path) as f:
                for name, rec in json.load(f)["kernels"].items():
                    if name.startswith(kernel_prefix):
                        best, src = rec["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
        except Exception:
            pass
    return best, src


def cpu_baseline(cfg, targets, preds, norm, train, val, epochs, lr, budget_s=20.0):
    """The reference's CPU path restated ("port"), timed on this box's host cores on
    a bounded sample of the SAME workload and extrapolated linearly to the full step.  Two ports
    are timed and the FASTER one is reported: oracle/np_port.py (OpenBLAS sgemm per contraction,
    sub-nets on a thread pool -- the shape of Keras/TF-on-CPU) and oracle/dimo.c (plain loops,
    OpenMP)."""
    import subprocess
    from oracle.dimo import OracleEngine
    from oracle.np_port import NumpyPort
    cores = os.cpu_count() or 1
    K, B, n = targets.shape[0], cfg["B"], norm.shape[0]
    steps_per_epoch = -(-train.size // B)
    n_sample = min(n, 4096)
    sub = np.ascontiguousarray(norm[:n_sample])         # the sampled cells; all genes
    notes = []

    def timed_steps(step_fn, budget):
        step_fn(0)                                       # warm-up
        t0 = time.time(); i = 0
        while i < 2 or (time.time() - t0 < budget and i < 64):
            i += 1; step_fn(i)
        return (time.time() - t0) / i, i

    rows_of = lambda i: (np.arange(B, dtype=np.int32) + (i * B) % (n_sample - B)).astype(np.int32)
    best = None
    tried = []                # every configuration that was timed: threads used -> extrapolated cells/s (all physical cores is one of them)
    # --- BLAS port, small search over (concurrent sub-nets) x (BLAS threads) ---
    try:
        from threadpoolctl import threadpool_limits
        port = NumpyPort([len(p) for p in preds], cfg["H"], cfg["O"], batch_size=B, dropout_rate=0.2, learning_rate=lr,
                         threads=min(K, cores))
        port.set_matrix(sub)
        for k in range(K):
            port.set_indices(k, preds[k], targets[k])
        port.gather(True)
        port.init_weights()
        for blas in sorted({1, max(1, cores // (2 * K)), max(1, cores // K)}):
            with threadpool_limits(limits=blas, user_api="blas"):
                t_step, cnt = timed_steps(lambda i: port.train_step(rows_of(i)), budget_s * 0.15)
                t0 = time.time(); port.predict(np.arange(512)); t_row = (time.time() - t0) / 512
            notes.append("np_port[blas=%d x pool=%d]: %.4f s/step, %.2e s/row" % (blas, min(K, cores), t_step, t_row))
            tried.append({"port": "np_port", "threads": blas * min(K, cores), "cells_per_s": n / (epochs * (steps_per_epoch * t_step + val.size * t_row) + n * t_row)})
            if best is None or t_step < best[0]:
                best = (t_step, t_row, "oracle/np_port.py (OpenBLAS, %d BLAS threads x %d concurrent sub-nets)" % (blas, min(K, cores)), cnt,
                        blas * min(K, cores))
        port.close()
    except Exception as e:
        notes.append("np_port failed: %r" % (e,))
    # --- plain-loop C oracle, rebuilt for this host's ISA ---
    lib = "/tmp/libdimo_native_%d.so" % os.getpid()
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-std=gnu11",
                               "-ffp-contract=off", "-o", lib, os.path.join(ROOT, "oracle", "dimo.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        lib = None
    try:
        eng = make_engine(OracleEngine, cfg, targets, preds, sub, np.arange(n_sample - 64, dtype=np.int32),
                          np.arange(n_sample - 64, n_sample, dtype=np.int32), [K], [0], 0, 0, lr, lib_path=lib)
        eng.init_weights()
        t_step, cnt = timed_steps(lambda i: eng.train_step(rows_of(i), epoch_key=0, step_key=i), budget_s * 0.3)
        t0 = time.time(); eng.predict(np.arange(256, dtype=np.int32)); t_row = (time.time() - t0) / 256
        notes.append("dimo.c[OpenMP %d threads]: %.4f s/step, %.2e s/row" % (cores, t_step, t_row))
        tried.append({"port": "dimo.c", "threads": cores, "cells_per_s": n / (epochs * (steps_per_epoch * t_step + val.size * t_row) + n * t_row)})
        if best is None or t_step < best[0]:
            best = (t_step, t_row, "oracle/dimo.c (OpenMP, %d threads)" % cores, cnt, cores)
        eng.close()
    except Exception as e:
        notes.append("dimo.c failed: %r" % (e,))
    if lib and os.path.exists(lib):
        os.remove(lib)
    if best is None:
        raise RuntimeError("; ".join(notes))
    t_step, t_row, which, cnt, threads = best
    t_full = epochs * (steps_per_epoch * t_step + val.size * t_row) + n * t_row
    fastest = n / t_full
    # the CPU path is quoted with ALL host cores busy (OMP_NUM_THREADS = core count).  `value` is therefore the
    # configuration that used the most host threads; the faster configuration this search found (fewer threads: the sub-nets' GEMMs are
    # too small for 256 of them) stands beside it as `fastest_value` / `fastest_threads`.  Both are "port" figures of the restated path
    # on a sample, extrapolated linearly -- a reported baseline, never Keras.
    most = max(tried, key=lambda r: (r["threads"], r["cells_per_s"]))
    return {"value": most["cells_per_s"], "unit": "cells/s", "cores": min(most["threads"], cores), "kind": "port", "host_threads_available": cores,
            "all_cores_value": most["cells_per_s"], "all_cores_threads": most["threads"], "all_cores_port": most["port"],
            "fastest_value": fastest, "fastest_threads": min(threads, cores), "fastest_port": which,
            "configurations": tried,
            "sample": "value: the configuration with the most host threads (%s, %d threads; BASELINE.md section 3); fastest: %s: %d train steps at %.4f s/step "
                      "+ forward at %.2e s/row on a %d-cell sample, extrapolated to %d epochs x %d steps + validation + predict of %d cells. All timings: %s"
                      % (most["port"], most["threads"], which, cnt, t_step, t_row, n_sample, epochs, steps_per_epoch, n, "; ".join(notes))}


def dropin_run(norm, epochs):
    """MultiNet(max_epochs=E).fit(raw) + predict(raw) on the counts the bench's log1p matrix came from."""
    import contextlib
    import io
    import pandas as pd
    from deepimpute_amd.multinet import MultiNet
    n, g = norm.shape
    raw = pd.DataFrame(np.rint(np.expm1(norm.astype(np.float64))), index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
    with contextlib.redirect_stdout(io.StringIO()):                       # (the reference's messages: stdout carries one JSON line)
        # Untimed warm-up of the drop-in surface on a corner of the matrix (one epoch, 512 cells x 1024 genes): the engine leg had its warm-up
        # imputes, this leg's first call otherwise pays for every lazy import and dlopen (pandas / scipy sub-modules, libhdf5 and its
        # dependencies) from a fresh box's cold file system -- 2-15 s inside fit() on the round's loaded boxes (profiles/r04_bench_other_boxes_summary.txt)
        t_w = time.perf_counter()
        corner = raw.iloc[:512, :1024]
        warm = MultiNet(verbose=0, max_epochs=1, patience=10 ** 6)
        warm.fit(corner, NN_lim=1024)
        warm.predict(corner)
        warm.close(release_cache=False)          # (a long-lived process: the blocks the engine leg used stay in the library's cache for the timed fit)
        del warm, corner
        warmup_s = time.perf_counter() - t_w
        net = MultiNet(verbose=0, max_epochs=epochs, patience=10 ** 6)     # the bench fixes E epochs on every leg
        t0 = time.perf_counter()
        net.fit(raw, NN_lim=g)
        t1 = time.perf_counter()
        out = net.predict(raw)
        t2 = time.perf_counter()
    assert out.shape == raw.shape and net.trained_epochs == epochs
    metrics = {k: float(v) for k, v in (net.test_metrics or {}).items()}      # multinet.py:251-262: Pearson r / MSE on the held-out cells' positive targets
    stages = {k: round(float(v), 4) for k, v in getattr(net, "timings", {}).items()}
    net.close()
    return {"fit_s": t1 - t0, "predict_s": t2 - t1, "cells_per_s": n / (t2 - t0), "subnets": len(net.predictors), "epochs": int(net.trained_epochs),
            # what the first call of a process pays on top (lazy imports, dlopen, the epilogue's device blocks): the untimed warm-up counted in;
            # the CLI as one cold process, CSV edges included: profiles/r05_cli_cold.txt
            "cells_per_s_with_warmup": n / (t2 - t0 + warmup_s),
            "test_metrics": metrics, "stages_s": stages, "warmup_s": round(warmup_s, 3),
            "note": "MultiNet.fit + predict on the same matrix as raw counts: host planning, host<->device copies of the counts and of the "
                    "imputed frame included; warmup_s: an untimed one-epoch fit + predict on a 512 x 1024 corner first (lazy imports, dlopen)"}


def deviation(a, b):
    """Element-wise deviation of `a` from `b` (two prediction matrices of the same problem): the figures north_star's "imputed values
    within 1e-4 relative" is about.  rel = |a - b| / |b| (softplus outputs: b > 0); `outside_tolerance` is the fraction of elements
    that miss the parity tests' criterion |a - b| <= 1e-4 |b| + 1e-5."""
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    d = np.abs(a - b)
    rel = d / np.maximum(np.abs(b), 1e-30)
    return {"max_rel": float(rel.max()), "p999_rel": float(np.quantile(rel, 0.999)), "median_rel": float(np.median(rel)),
            "rms_rel": float(np.sqrt(np.mean(rel * rel))), "max_abs": float(d.max()),
            "outside_tolerance": float(np.mean(d > 1e-4 * np.abs(b) + 1e-5)), "elements": int(a.size)}


def accuracy_pair(cfg, targets, preds, norm, epochs, lr, n_cells=4096, n_subnets=4):
    """The accuracy half of the metric ("MSE vs ref"; north_star: "imputed values within 1e-4 relative"), at the horizon the bench
    times: after the SAME E epochs of the SAME problem -- the first `n_cells` cells, 5 % held out, same seeds and Philox streams --
    (1) the HIP engine with ALL K sub-nets of the config, i.e. on exactly the kernels, work tables and split-K partitions of the
        timed job (only the cell count differs), and, as `hip_share`, with the first five sub-nets alone (one rank of the 8-GPU
        job: the register-resident kernel);
    (2) the CPU port (oracle/dimo.c, the restatement of the reference's Keras path) in float32 and in float64 on the first
        `n_subnets` sub-nets (its cost is per sub-net; the Philox streams are keyed by global sub-net, so these are the same
        sub-nets).
    Reported: the held-out metrics fit() computes (multinet.py:251-262) from each, and the ELEMENT-WISE deviation of the predicted
    matrix over all sample cells (model.predict: the log1p-space values, multinet.py:278-280; and their expm1, the imputed counts of
    :294) after 1, 3 and E epochs -- HIP vs the float32 port, and each of them vs the float64 port: the distance of the float32 port
    from the float64 one is the noise floor ANY float32 evaluation of that many optimiser steps has (a pre-activation that is zero to
    float32 precision takes its relu gate on either side, Adam turns the changed gradient into a full-size step: DESIGN section 4).
    `within_noise_floor`: the HIP path is no further from the float64 trajectory than twice the float32 port is (rms).
    Outside the timed region."""
    import ctypes
    from deepimpute_amd.engine import HipEngine
    from oracle.dimo import OracleEngine
    try:        # the oracle's OpenMP regions are tiny: on a 256-thread host fork/join dominates (250 s instead of ~15 s for this leg)
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(min(32, os.cpu_count() or 1))
    except OSError:
        pass
    K = targets.shape[0]
    n_subnets = min(n_subnets, K)
    n_cells = min(n_cells, norm.shape[0])
    sub = np.ascontiguousarray(norm[:n_cells])
    train, val = split_rows(n_cells, seed=0)
    O = cfg["O"]
    k_all = min(n_subnets, 5, K)                                                # the sub-nets all four runs share
    checkpoints = sorted({e for e in (1, 3, epochs) if e <= epochs})
    runs = (("hip", HipEngine, K, {}), ("hip_share", HipEngine, min(5, K), {}),
            ("cpu_port", OracleEngine, n_subnets, {}), ("cpu_port_fp64", OracleEngine, n_subnets, {"fp64": True}))
    out, pred = {}, {}
    for name, cls, k_run, kw in runs:
        t0 = time.time()
        eng = make_engine(cls, cfg, targets[:k_run], preds[:k_run], sub, train, val, [k_run], [0], 0, 0, lr, **kw)
        eng.gather(True)
        eng.init_weights()
        pred[name] = {}
        for e in range(epochs):
            eng.train_epoch(e)
            if e + 1 in checkpoints:
                pred[name][e + 1] = eng.predict()[:, :k_all * O].copy()         # every sample cell
        k_cmp = min(k_run, n_subnets)
        vl = float(np.sum(eng.val_loss()[:k_cmp]))
        guess = pred[name][epochs][val].reshape(-1).astype(np.float64)
        truth = np.hstack([sub[np.ix_(val, targets[k])] for k in range(k_all)]).reshape(-1).astype(np.float64)
        pos = truth > 0
        truth, guess = truth[pos], guess[pos]
        out[name] = {"correlation": float(np.corrcoef(truth, guess)[0, 1]), "MSE": float(np.mean((truth - guess) ** 2)), "val_loss": vl,
                     "subnets_trained": k_run, "seconds": time.time() - t0}
        if hasattr(eng, "path_info"):
            out[name]["path"] = eng.path_info()
        eng.close()
    out["relative_difference"] = {k: abs(out["hip"][k] - out["cpu_port"][k]) / abs(out["cpu_port"][k]) for k in ("correlation", "MSE", "val_loss")}
    out["relative_difference_vs_fp64"] = {who: {k: abs(out[who][k] - out["cpu_port_fp64"][k]) / abs(out["cpu_port_fp64"][k]) for k in ("correlation", "MSE", "val_loss")}
                                          for who in ("hip", "cpu_port")}
    pairs = (("hip_vs_cpu_port", "hip", "cpu_port"), ("hip_vs_cpu_port_fp64", "hip", "cpu_port_fp64"),
             ("hip_share_vs_cpu_port_fp64", "hip_share", "cpu_port_fp64"), ("cpu_port_fp32_vs_fp64", "cpu_port", "cpu_port_fp64"))
    final = {label: deviation(pred[x][epochs], pred[y][epochs]) for label, x, y in pairs}
    brief = lambda d: {k: d[k] for k in ("max_rel", "p999_rel", "rms_rel", "outside_tolerance")}
    by_epoch = {str(e): {label: brief(deviation(pred[x][e], pred[y][e])) for label, x, y in pairs} for e in checkpoints}
    per_subnet = [{label: deviation(pred[x][epochs][:, k * O:(k + 1) * O], pred[y][epochs][:, k * O:(k + 1) * O])["rms_rel"] for label, x, y in pairs}
                  for k in range(k_all)]
    ex = lambda name: np.expm1(pred[name][epochs].astype(np.float64))
    out["imputed_values"] = {
        "log1p_space": final, "by_epoch": by_epoch, "rms_rel_per_subnet": per_subnet,
        "counts_expm1": {"hip_vs_cpu_port": deviation(ex("hip"), ex("cpu_port")), "hip_vs_cpu_port_fp64": deviation(ex("hip"), ex("cpu_port_fp64")),
                         "cpu_port_fp32_vs_fp64": deviation(ex("cpu_port"), ex("cpu_port_fp64"))},
        "within_noise_floor": bool(final["hip_vs_cpu_port_fp64"]["rms_rel"] <= 2.0 * final["cpu_port_fp32_vs_fp64"]["rms_rel"] + 1e-7),
        "note": "element-wise over the predictions of ALL %d sample cells x %d sub-nets x %d targets; by_epoch: after 1 / 3 / %d epochs of %d optimiser "
                "steps; rel = |a - b| / |b|; outside_tolerance = share of elements beyond 1e-4 |b| + 1e-5 (the parity tests' criterion). "
                "cpu_port_fp32_vs_fp64 is the float32 noise floor of the same trajectory: once a float32 path takes a relu gate of a pre-activation "
                "at rounding level on the other side, that sub-net's values move by 1e-3 .. 1e-1 relative -- in EITHER float32 path (DESIGN section 4)"
                % (n_cells, k_all, O, epochs, -(-train.size // cfg["B"]))}
    out["sample"] = "%d cells (5 %% held out), %d epochs, same seeds / Philox streams everywhere; hip: all %d sub-nets (the timed job's kernels), " \
                    "hip_share: %d (resident kernel), cpu ports: %d; compared on the first %d" % (n_cells, epochs, K, min(5, K), n_subnets, k_all)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--epochs", type=int, default=18, help="fixed epoch count E of every fit (see DESIGN.md)")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in MultiNet.fit+predict run (config.dropin)")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the HIP vs CPU-port held-out metrics after E epochs on a sub-problem (accuracy)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--accuracy-cells", type=int, default=4096, help="cells of the accuracy leg's problem (the bench's own 50000 for the full horizon: the CPU oracles then take minutes)")
    ap.add_argument("--accuracy-subnets", type=int, default=4, help="sub-nets the CPU oracles of the accuracy leg run")
    ap.add_argument("--limit-subnets", type=int, default=0, help="diagnostic: keep only the first N sub-nets (what one rank of an N-GPU job sees)")
    ap.add_argument("--hidden", type=int, default=0, help="diagnostic: hidden width (default: the config's 256; the reference CLI defaults to 300)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"], help="bf16: X arena in bfloat16, inference / validation on the bf16 matrix cores")
    ap.add_argument("--stream", action="store_true", help="stream the matrix from host memory in row blocks (it never resides on the GPU)")
    ap.add_argument("--cells", type=int, default=0, help="diagnostic: override the config's cell count")
    ap.add_argument("--batch", type=int, default=0, help="diagnostic: batch size (default: the config's 64; above 64 the general path runs, as for deepImpute --batch-size 128)")
    ap.add_argument("--general", action="store_true", help="diagnostic: force the general path (dimn_create_general) at the config's shapes")
    ap.add_argument("--early-stop-probe", action="store_true", help="also run the early-stopped fit once and report its epoch count")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DIMN_BENCH_DEVICE"):                # diagnostic: several ranks on one GPU (plumbing check only)
        local_rank = int(os.environ["DIMN_BENCH_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    from deepimpute_amd.engine import HipEngine
    if world == 1:                                  # what a long-lived process does at start-up: HIP context + pinned bounce buffers, on a helper thread
        from deepimpute_amd import _lib
        _lib.warm_up_async(local_rank)
    cfg = dict(CONFIGS[args.config])
    if args.hidden:
        cfg["H"] = args.hidden
        cfg["label"] += " [hidden=%d]" % args.hidden
    if args.cells:
        cfg["n"] = args.cells
        cfg["label"] += " [cells=%d]" % args.cells
    if args.batch:
        cfg["B"] = args.batch
        cfg["label"] += " [batch=%d]" % args.batch
    general = args.general or cfg["B"] > 64 or cfg["H"] > 384
    if general:
        cfg["label"] += " [general path]"
    n, g = cfg["n"], cfg["g"]
    t_gen = time.time()
    rdzv = FileRendezvous(rank, world) if world > 1 else None
    shared_seg = None
    if world > 1 and args.stream:
        norm, shared_seg = shared_matrix(rdzv, n, g, seed=0)     # one host copy for the node's ranks; each rank's hand-over starts elsewhere in it
    else:
        norm = synth_counts(n, g, seed=0)
    targets, preds = synth_indices(g, cfg["O"], seed=0)
    train, val = split_rows(n, seed=0)
    if args.limit_subnets:
        targets, preds = targets[:args.limit_subnets], preds[:args.limit_subnets]
    K = targets.shape[0]
    counts, offs = shard(K, world)
    t_gen = time.time() - t_gen

    if general:
        from deepimpute_amd.engine import HipGeneralEngine

        def general_factory(D, hidden, out_dim, dropout_rate=0.2, **kw):      # the one-hidden-layer model through dimn_create_general
            return HipGeneralEngine(D, [(hidden, "relu", dropout_rate)], out_dim, **kw)
        engine_cls = general_factory
    else:
        engine_cls = HipEngine
    def build_engine():
        return make_engine(engine_cls, cfg, targets, preds, norm, train, val, counts, offs, rank, local_rank, args.lr, stream=args.stream,
                           **({"precision": "bf16"} if args.precision == "bf16" else {}))
    comm = None
    if world > 1:
        eng, comm, rccl_error = bring_up_job(build_engine, rdzv, rank, world)
        if comm is None:
            sys.stderr.write("bench.py rank %d: the %d-rank job did not come up (%s): no measurement\n" % (rank, world, rccl_error))
            if rank == 0:
                print(json.dumps(rccl_failure_line(args, world, cfg["label"], rccl_error)))
                sys.stdout.flush()
            file_vote(rdzv, "bye", 0.0, timeout=60.0)            # nobody removes the directory under a rank still reading it
            rdzv.cleanup()
            if eng is not None:
                eng.close()
            sys.exit(3)
    else:
        eng = build_engine()

    def barrier():
        eng.synchronize()
        if comm:
            comm.allreduce_sum(np.zeros(1), timed=False)

    for _ in range(args.warmup):
        impute_once(eng, args.epochs, comm, counts, n)
    eng.set_profiling(True)
    eng.get_timers(reset=True)
    eng._bench_train_s, eng._bench_train_steps = 0.0, 0
    if comm:
        comm.reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vsum = impute_once(eng, args.epochs, comm, counts, n)
    barrier()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)
    timers = eng.get_timers(reset=True)
    per_rank = None
    if comm:
        # one row per rank, so that a multi-GPU line explains itself: wall time of the timed region, time per optimiser step on
        # that rank's own stream (HIP events), its share of sub-nets, the all-reduce per epoch, predict, and the gather
        mine = np.zeros((world, 7))
        mine[rank] = [dt, (timers[6] / timers[7]) if timers[7] > 0 else timers[0] / max(1.0, timers[1]), counts[rank],
                      1e3 * comm.allreduce_s / max(1, comm.allreduces), 1e3 * getattr(eng, "_bench_predict_s", 0.0),
                      1e3 * comm.gather_s / max(1, comm.gathers), float(timers[7] > 0)]
        allr = comm.allreduce_sum(mine.ravel(), timed=False).reshape(world, 7)
        dt = float(allr[:, 0].max())
        nranks, _ = eng.comm_info()
        per_rank = {"nranks_ncclCommCount": nranks, "wall_s": [float(x) for x in allr[:, 0]], "lane_step_ms": [float(x) for x in allr[:, 1]],
                    "subnets": [int(x) for x in allr[:, 2]], "allreduce_ms_per_epoch": [float(x) for x in allr[:, 3]],
                    "predict_ms": [float(x) for x in allr[:, 4]], "gather_ms": [float(x) for x in allr[:, 5]],
                    "resident_kernel": [bool(x) for x in allr[:, 6]],
                    "gather_bytes_into_root": int(comm.gather_bytes // max(1, comm.gathers)),
                    "gather_GBps_into_root": (comm.gather_bytes / comm.gather_s / 1e9) if comm.gather_s > 0 else None}

    result = None
    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = n / (dt / args.steps)
        # dominant kernel: W1 gradient + Adam + next forward.  libdimn brackets every launch (one per
        # sub-net lane and step) with HIP events on the launch's own stream and sums its ALGORITHMIC
        # bytes (24 B per W1 parameter + X tiles + dA)
        w1_ms = timers[2] / max(1.0, timers[3])
        abytes = timers[4] / max(1.0, timers[3])
        achieved = timers[4] / (timers[2] * 1e-3) / 1e9 if timers[2] > 0 else 0.0
        steps_per_epoch = -(-train.size // cfg["B"])
        # whole-job ALGORITHMIC flops: training 4*D*H + 6*H*O per sample and sub-net
        # (forward + dW1 + dW2 + dD), forward 2*D*H + 2*H*O; E epochs of train + validation, then predict
        H_, O_ = cfg["H"], cfg["O"]
        f_tr = sum(4.0 * len(p) * H_ + 6.0 * H_ * O_ for p in preds)
        f_fw = sum(2.0 * len(p) * H_ + 2.0 * H_ * O_ for p in preds)
        job_flops = args.epochs * (train.size * f_tr + val.size * f_fw) + n * f_fw
        job_tflops = job_flops / (dt / args.steps) / 1e12
        job_mfma = {"achieved": job_tflops, "peak": F32_MFMA_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                    "frac": job_tflops / (F32_MFMA_PEAK_TFLOPS * world), "algorithmic_flops": job_flops}
        if timers[7] > 0:
            # few sub-nets per GPU: the register-resident epoch kernel (dimn_resident.h) ran -- the optimiser state never
            # leaves the register file, so the kernel is bounded by the fp32 matrix pipe and by the hand-offs between
            # workgroups, not by HBM: achieved = ALGORITHMIC training flops of this rank's sub-nets per optimiser step
            # (4*D*H + 6*H*O per sample and sub-net) / the launch's time per step (HIP events)
            lane_step_ms = timers[6] / timers[7]
            mine = preds[offs[0]:offs[0] + counts[0]]
            step_flops = cfg["B"] * sum(4.0 * len(p) * H_ + 6.0 * H_ * O_ for p in mine)
            ach = step_flops / (lane_step_ms * 1e-3) / 1e12
            # (a rank whose sub-nets do not fit the register file at once trains them in up to three groups, one launch each per
            #  epoch: avg_launch_ms is then the epoch's launches together, steps_per_launch the optimiser steps they cover)
            roofline = {"bound": "mfma", "kernel": "k_epoch_resident (one launch per epoch and group of <= 5 sub-nets: W, m, v of both layers in registers/LDS)",
                        "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK_TFLOPS,
                        "traffic": None, "algorithmic_flops_per_step": step_flops, "avg_launch_ms": timers[6] / max(1, args.steps * args.epochs),
                        "launches": args.steps * args.epochs, "steps_per_launch": steps_per_epoch, "job_mfma": job_mfma}
        else:
            lane_step_ms = timers[0] / max(1.0, timers[1])
            traffic, traffic_src = measured_traffic("k_w1_update_fwd")
            try:
                fl = eng.path_info().get("first_layer", 1)
            except Exception:
                fl = 1
            b1f1 = {1: "k_w1_update_fwd_ring<16,1>", 2: "k_w1_update_fwd_sh<10,2>", 3: "k_w1_update_fwd_ring<waves,1,4> (one hidden tile per wave, four-set ring)"}.get(fl, "k_w1_update_fwd<NT2>")
            roofline = {"bound": "hbm", "kernel": b1f1 + " (W1 grad + Adam + next forward)", "achieved": achieved,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        # PMC figure of the SAME command from a committed profile (separate rocprofv3 --pmc passes cannot run
                        # inside this process); traffic_source names the file it was read from
                        "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": w1_ms, "launches": int(timers[3]),
                        # secondary view the north star asks for: the whole job against the dense fp32-MFMA peak.
                        # At batch 64 a training step has ~9-14 flop per byte of weight + Adam traffic, far below the
                        # ~20 flop/B ridge of fp32 MFMA vs HBM, so this fraction is bounded by the HBM figure above.
                        "job_mfma": job_mfma}
            # the WHOLE optimiser step against the HBM roof: 28 B per parameter of both layers (forward read + Adam
            # read-modify-write of w, m, v) + the batch rows of X + the targets, over lane_step_ms (HIP events around whole steps)
            step_bytes = sum(28.0 * (len(p) * H_ + H_ * O_) + 4.0 * cfg["B"] * (len(p) + O_) for p in preds[offs[0]:offs[0] + counts[0]])
            roofline["step_algorithmic_bytes"] = step_bytes
            roofline["step_frac"] = step_bytes / (lane_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if lane_step_ms > 0 else None
        # scalars beside the nested records (the driver's record keeps scalars of `config` / `roofline` / `cpu_baseline` and drops nested dicts)
        roofline["job_mfma_frac"] = job_mfma["frac"]
        result = {
            "metric": "cells/sec end-to-end impute (fit+predict)", "value": value, "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16 (X arena, inference GEMMs; fp32 accumulate) + f32 (weights, Adam, training GEMMs)",
            "data": "synthetic (seeded Poisson-Gamma counts, BASELINE.md generator; random-init Glorot weights)",
            "config": {"workload": cfg["label"], "cells": n, "genes": g, "subnets": K, "epochs_per_fit": args.epochs,
                       "train_steps_per_epoch": steps_per_epoch, "parallelism": "subnets sharded x%d" % world,
                       "final_val_loss": vsum, "subnet_lanes": int(timers[5]), "matrix": "streamed from host (pinned row blocks)" if args.stream else "resident",
                       "lane_step_ms": lane_step_ms,
                       # `value` is the ENGINE figure: the matrix resident in HBM when the timed region starts, the predictions left in HBM, no
                       # planning, no PCIe (the bench contract); the figure BASELINE's metric wording describes -- MultiNet.fit + predict from a
                       # host frame to a host frame -- is config.dropin.cells_per_s.  The timed region carries the HIP-event stamps of one
                       # launch in eight (set_profiling): they make `value` conservative, never optimistic.
                       "value_scope": "engine: resident matrix -> predictions in HBM; host-to-host figure: config.dropin; event stamps inside the timed region",
                       # host wall time of the train_epoch calls per optimiser step (every path, also the general one, which has no event timers)
                       "train_step_ms_wall": 1e3 * eng._bench_train_s / max(1, eng._bench_train_steps)},
            "roofline": roofline,
        }
        if getattr(eng, "_bench_handover_s", None):
            result["config"]["streamed_handover"] = {"seconds": eng._bench_handover_s, "matrix_GB": norm.nbytes / 1e9,
                                                    "host_to_device_GBps": norm.nbytes / 1e9 / eng._bench_handover_s,
                                                    "note": "last impute: pageable host matrix -> pinned bounce buffers (threads) -> PCIe -> device gather, per ~128 MB row block"}
        # the forward over all cells (model.predict), the MFMA-bound kernel of the path: fp32 matrix cores, or bf16 ones with --precision bf16
        pred_s = getattr(eng, "_bench_predict_s", None)
        if pred_s:
            mine = preds[offs[0]:offs[0] + counts[0]]
            pf = n * sum(2.0 * len(p) * H_ + 2.0 * H_ * O_ for p in mine)
            if args.precision == "bf16":
                # 217 flop per byte: BELOW the bf16 ridge (2.5 PFLOP/s / 8 TB/s = 312), so the forward on the bf16 matrix cores is bounded by
                # HBM: algorithmic bytes = the bf16 predictor blocks once + the fp32 predictions written once
                pb = n * sum(2.0 * len(p) + 4.0 * O_ for p in mine)
                result["roofline"]["predict"] = {"kernel": "k_predict_bf16", "bound": "hbm", "ms": 1e3 * pred_s, "achieved": pb / pred_s / 1e9, "peak": HBM_PEAK_GBS,
                                                 "unit": "GB/s", "frac": pb / pred_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": pb,
                                                 "mfma": {"achieved": pf / pred_s / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": pf / pred_s / 1e12 / 2500.0}}
            else:
                result["roofline"]["predict"] = {"kernel": "k_predict", "bound": "mfma", "ms": 1e3 * pred_s, "achieved": pf / pred_s / 1e12,
                                                 "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": pf / pred_s / 1e12 / F32_MFMA_PEAK_TFLOPS}
        if pred_s and "predict" in result["roofline"]:
            result["roofline"]["predict_frac"] = result["roofline"]["predict"]["frac"]
            result["roofline"]["predict_ms"] = result["roofline"]["predict"]["ms"]
            result["roofline"]["predict_bound"] = result["roofline"]["predict"]["bound"]
        if args.early_stop_probe:
            eng.gather(True); eng.init_weights()
            t1 = time.perf_counter()
            ne, lh, vh = eng.fit(500, 5)
            eng.predict_device(); eng.synchronize()
            result["config"]["early_stopped"] = {"epochs": int(ne), "seconds": time.perf_counter() - t1,
                                                 "val_loss": [float(x) for x in vh[-6:]]}
    if comm:
        if rank == 0 and result is not None:
            result["config"]["collectives"] = "rccl"
            result["config"]["per_rank"] = per_rank
        comm.allreduce_sum(np.zeros(1), timed=False)
        comm.close()
        rdzv.cleanup()
    eng.close()
    if rank == 0 and world == 1 and not args.no_dropin and not args.limit_subnets and not args.hidden and not general:
        # The drop-in surface on the SAME matrix, outside the timed region: deepimpute_amd.multinet.MultiNet.fit + predict,
        # host planning included (gene selection, |corr| + predictor selection, split, save, held-out metrics, post-processing;
        # raw counts and the returned frame live on the host, so this figure includes the PCIe copies `value` excludes)
        try:
            d = result["config"]["dropin"] = dropin_run(norm, args.epochs)
            # the figure BASELINE's metric wording describes (host frame -> imputed host frame), as scalars the driver's record keeps
            result["config"].update({"dropin_cells_per_s": d["cells_per_s"], "dropin_fit_s": d["fit_s"], "dropin_predict_s": d["predict_s"],
                                     "dropin_cells_per_s_with_warmup": d["cells_per_s_with_warmup"], "dropin_train_s": d["stages_s"].get("fit.train")})
        except Exception as e:
            result["config"]["dropin"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_accuracy and not args.limit_subnets and not general and args.precision == "fp32":
        try:
            result["accuracy"] = accuracy_pair(cfg, targets, preds, norm, args.epochs, args.lr, n_cells=args.accuracy_cells, n_subnets=args.accuracy_subnets)
            iv = result["accuracy"]["imputed_values"]
            for e_, rec in iv["by_epoch"].items():          # share of imputed values beyond 1e-4 |b| + 1e-5 after 1 / 3 / E epochs, HIP vs the float32 and float64 ports
                tag = "E" if int(e_) == args.epochs else e_
                result["config"]["accuracy_outside_tol_epoch%s" % tag] = rec["hip_vs_cpu_port"]["outside_tolerance"]
                result["config"]["accuracy_outside_tol_vs_fp64_epoch%s" % tag] = rec["hip_vs_cpu_port_fp64"]["outside_tolerance"]
                result["config"]["accuracy_fp32_port_outside_tol_vs_fp64_epoch%s" % tag] = rec["cpu_port_fp32_vs_fp64"]["outside_tolerance"]
            result["config"]["accuracy_within_noise_floor"] = iv["within_noise_floor"]
            result["config"]["accuracy_rms_rel_hip_vs_fp64"] = iv["log1p_space"]["hip_vs_cpu_port_fp64"]["rms_rel"]
            result["config"]["accuracy_rms_rel_fp32_port_vs_fp64"] = iv["log1p_space"]["cpu_port_fp32_vs_fp64"]["rms_rel"]
            result["config"]["accuracy_val_loss_rel_diff_vs_fp64"] = result["accuracy"]["relative_difference_vs_fp64"]["hip"]["val_loss"]
        except Exception as e:
            result["accuracy"] = {"error": repr(e)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(cfg, targets, preds, norm, train, val, args.epochs, args.lr, args.cpu_budget)
            except Exception as e:   # the baseline is a reported figure, never the product path
                result["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        else:
            result["cpu_baseline"] = None
        result["config"]["synth_seconds"] = t_gen
        print(json.dumps(result))


if __name__ == "__main__":
    main()
