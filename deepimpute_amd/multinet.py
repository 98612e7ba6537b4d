"""Drop-in `MultiNet` estimator on the MI355X-native engine.

Public surface = the reference's `deepimpute.multinet`: class `MultiNet` (fit / predict /
score, attributes NN_parameters, predictors, targets, trained_epochs, test_metrics ...) and the
module functions `get_distance_matrix`, `wMSE`, `inspect_data` (reference
deepimpute/multinet.py:20-63, 65-374).

What differs is what sits behind it.  The reference builds one Keras model with K inputs and
K outputs and calls model.fit / model.predict (multinet.py:126-167, 238-253, 276-280); here
those calls go to `deepimpute_amd.engine.HipEngine`, i.e. hand-written gfx950 kernels behind the
C ABI of include/dimn.h.  The shared log1p matrix is uploaded once and the per-sub-net
(cells x predictors)/(cells x targets) blocks are gathered on the GPU from column index lists,
instead of 4K pandas `.loc` copies on the host.

The host-side planning around the seam (which genes, which targets per sub-net, which
predictors, the 5 % validation split, the post-processing of predictions) follows the
reference's behaviour including its use of the global numpy RNG, so that a given seed yields the
same plan; tests/golden/shell_*.npz (captured from the imported reference) pin that.

There is no CPU fallback: without libdimn.so and a GPU, fit()/predict() raise.
"""
import glob
import json
import os
import re
import sys
import tempfile
import warnings

import numpy as np
import pandas as pd

from . import _hostpar

# one scratch directory per process, shared by every instance that does not pass its own
# (the reference evaluates tempfile.mkdtemp() once, as a default argument: multinet.py:74)
_SCRATCH = tempfile.mkdtemp()
_VALIDATION_FRACTION = 0.05     # multinet.py:228
_POST_ROWS = 256                # cells per host-pool task in predict()'s post-processing


# --------------------------------------------------------------------------- module functions
def _abs_corrcoef(values, backend="auto", device_id=0):
    """|np.corrcoef| of the columns of `values` ([cells, genes] float64).  backend "hip": fp64 MFMA
    kernel of libdimn (dimn_abs_corrcoef); "numpy": the reference's own host computation; "auto": hip
    when a GPU is visible, numpy otherwise (host planning is not part of the accelerated hot path, so
    unlike fit/predict it may run without a GPU)."""
    if backend not in ("auto", "hip", "numpy"):
        raise ValueError("backend must be 'auto', 'hip' or 'numpy'")
    if backend != "numpy":
        try:
            from . import _cabi, _lib
            fns = _lib.load()
            x = _hostpar.as_float64(values)
            out = np.empty((x.shape[1], x.shape[1]), np.float64)
            rc = fns["abs_corrcoef"](int(device_id), _cabi.p_f64(x), x.shape[0], x.shape[1], _cabi.p_f64(out))
            if rc == 0:
                return out
            msg = fns["last_error"]().decode("utf-8", "replace")
            # "auto" may run on the host only when NO GPU is visible; a failing GPU (out of memory, HIP error)
            # is an error, not a reason to spend minutes in float64 numpy without a word
            if backend == "hip" or "no HIP device visible" not in msg:
                raise RuntimeError("dimn_abs_corrcoef: " + msg)
        except ImportError:
            if backend == "hip":
                raise
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.abs(np.corrcoef(values.T))


def _candidate_pool(raw, n_pred=None, _var_mean=None, labels_only=False):
    """(labels, values) of the candidate predictor genes of get_distance_matrix (multinet.py:20-30): genes with
    std/mean > 0, or the `n_pred` genes with the largest ratio; values = their raw columns [cells, pool]
    (labels_only: None instead -- the caller reads the columns from the device-resident counts)."""
    var, mean = _var_mean if _var_mean is not None else _hostpar.column_var_mean(raw)
    ratio = np.sqrt(var) / mean
    ratio[np.isinf(ratio)] = 0
    if n_pred is None:
        keep = raw.columns[ratio > 0]
    else:
        print("Using {} predictors".format(n_pred))
        keep = ratio.sort_values(ascending=False).index[:n_pred]
    if labels_only:
        return keep, None
    if keep.equals(raw.columns):
        return keep, raw.values
    return keep, _hostpar.take_columns(raw.values, raw.columns.get_indexer(keep))


def get_distance_matrix(raw, n_pred=None, backend="auto", device_id=0, _var_mean=None):
    """Absolute Pearson correlation between candidate predictor genes (multinet.py:20-34).
    Candidates: genes with std/mean > 0, or the `n_pred` genes with the largest ratio.  The g x g
    float64 correlation itself runs on the GPU when one is visible (`backend`, see _abs_corrcoef).
    `_var_mean`: (raw.var(), raw.mean()) when the caller already holds them (std = sqrt(var), as in
    pandas)."""
    keep, candidates = _candidate_pool(raw, n_pred, _var_mean)
    corr = _abs_corrcoef(candidates, backend=backend, device_id=device_id)
    return pd.DataFrame(_hostpar.zero_nans_inplace(corr), index=keep, columns=keep, copy=False)   # .fillna(0)


class _ColumnsOnly:
    """What setTargets() reads of its `data` argument: `.columns` and `.shape[1]` (and, for fit()'s own bookkeeping when the
    log1p matrix lives on the device only, `.index` / `.shape`)."""

    def __init__(self, labels, index=None, shape=None):
        self.columns = pd.Index(labels)
        self.index = index
        self.shape = tuple(shape) if shape is not None else (0, len(self.columns))


def _gpu_visible():
    """True when libdimn loads and sees a HIP device (host planning may run without one; fit/predict may not)."""
    try:
        from . import _lib
        return _lib.device_count() > 0
    except (ImportError, OSError, AttributeError):
        return False


def wMSE(y_true, y_pred, binary=False):
    """The reference's loss (multinet.py:36-41) on numpy arrays: mean over every element of
    w * (y - yhat)^2 with w = y_true, or 1[y_true > 0] when `binary`.  Training evaluates it
    inside the kernels; this function exists for API compatibility and for checks."""
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    w = (y_true > 0).astype(np.float32) if binary else y_true
    return np.mean(w * np.square(y_true - y_pred))


def inspect_data(data, _max=None):
    """Guards of multinet.py:43-63: unique cell / gene labels, and raw (not log) counts (`_max`: the matrix maximum when the
    caller already holds it)."""
    problems = (("cell", data.index), ("gene", data.columns))
    for what, labels in problems:
        if sum(labels.duplicated()):
            print("ERROR: duplicated {0} labels. Please provide unique {0} labels.".format(what))
            exit(1)
    top = _hostpar.matrix_max(data.values) if _max is None else _max
    if top < 10:
        print("ERROR: max value = {}. Is your data log-transformed? Please provide raw counts".format(top))
        exit(1)
    print("Input dataset is {} cells (rows) and {} genes (columns)".format(*data.shape))
    print("First 3 rows and columns:")
    print(data.iloc[:3, :3])


def _parse_architecture(architecture):
    """[(neurons, activation, dropout rate behind it)] of an architecture list (multinet.py:135-143: a sequence of
    {"type": "dense", "neurons", "activation"} / {"type": "dropout", "rate"} entries; other types are skipped with the
    reference's message).  Consecutive Dropout layers compose (keep probabilities multiply would change the stream
    semantics, so they are rejected).  A Dropout before the first Dense layer (dropout on the inputs) becomes a leading
    (0, "linear", rate) entry: the general path masks the batch's predictor rows (dimn_create_general)."""
    from ._cabi import ACTIVATIONS
    layers = []
    for spec in architecture:
        kind = str(spec.get("type", "")).lower()
        if kind == "dense":
            name = spec.get("activation", None)
            act = "linear" if name is None else str(name).lower()
            if act not in ACTIVATIONS:
                raise NotImplementedError("hidden activation %r: the gfx950 kernels implement %s" % (name, sorted(ACTIVATIONS)))
            layers.append([int(spec["neurons"]), act, 0.0])
        elif kind == "dropout":
            if not layers:
                layers.append([0, "linear", float(spec["rate"])])          # dropout on the inputs
                continue
            if layers[-1][2] > 0.0:
                raise NotImplementedError("two Dropout layers in a row (got %r)" % (architecture,))
            layers[-1][2] = float(spec["rate"])
        else:
            print("Unknown layer type.")       # the reference skips such entries (multinet.py:142-143)
    if not [l for l in layers if l[0] > 0]:
        raise NotImplementedError("architecture needs at least one dense layer")
    return [tuple(l) for l in layers]


def _loss_name(loss):
    """The loss of NN_parameters['loss'] (multinet.py:150-162: the module's wMSE by name or callable, or a keras.losses
    name) as an id of the engines (_cabi.LOSSES: wmse / wmse_binary / mean_squared_error / mean_absolute_error / msle / logcosh / huber / poisson)."""
    from ._cabi import LOSSES
    if callable(loss):
        binary = bool(getattr(loss, "keywords", {}) and loss.keywords.get("binary"))       # functools.partial(wMSE, binary=True)
        loss = getattr(getattr(loss, "func", loss), "__name__", str(loss))
        if binary and str(loss).lower() == "wmse":
            loss = "wmse_binary"
    name = str(loss).lower()
    if name in LOSSES:
        return name
    print('Unknown loss: {}. Aborting.'.format(loss))       # multinet.py:160-161 (_cabi.LOSSES: wMSE and the element-wise keras.losses)
    exit(1)


class _Stages(dict):
    """Wall time per stage of the last fit() / predict() (extension: MultiNet.timings; bench.py reports it)."""

    def stage(self, name):
        import contextlib
        import time

        @contextlib.contextmanager
        def cm():
            t0 = time.perf_counter()
            try:
                yield
            finally:
                self[name] = self.get(name, 0.0) + time.perf_counter() - t0
        return cm()


_pending_saves = {}            # directory (real path) -> (thread, box): weight files a fit() of this process is still writing


def _start_save(directory, work):
    import threading
    key = os.path.realpath(directory)
    box = {}

    def run():
        try:
            work()
        except BaseException as exc:                     # re-raised where the files are needed
            box["error"] = exc
    thread = threading.Thread(target=run, name="dimn-save")      # not a daemon: the interpreter waits for the files at exit
    _pending_saves[key] = (thread, box)
    if not _pending_saves_hooked:
        import atexit
        _pending_saves_hooked.append(True)
        atexit.register(_report_unjoined_saves)
    thread.start()


_pending_saves_hooked = []


def _report_unjoined_saves():
    """atexit: a write nobody joined (fit() joins its own: only a raise between save() and the join leaves one) must not fail silently."""
    for key in list(_pending_saves):
        try:
            _join_saves(key)
        except BaseException as exc:
            sys.stderr.write("deepimpute_amd: writing the model files in %s failed: %r\n" % (key, exc))


def _join_saves(directory=None):
    """Wait until the weight files of `directory` (None: of every directory) are on disk; a failed write raises here."""
    keys = list(_pending_saves) if directory is None else [os.path.realpath(directory)]
    for key in keys:
        entry = _pending_saves.pop(key, None)
        if entry is not None:
            entry[0].join()
            if "error" in entry[1]:
                raise entry[1]["error"]


def _shard_rank(path):
    """r of .../model.rank<r>.npz, None for anything else."""
    m = re.fullmatch(r"model\.rank(\d+)\.npz", os.path.basename(path))
    return int(m.group(1)) if m else None


# ------------------------------------------------------------------------------- the estimator
class MultiNet:
    # planning on the device from ONE upload of the raw counts (fit(): gene statistics, correlation, predictor selection; predict():
    # restore / max against the resident counts); needs the HIP engine's dimn_set_matrix_counts
    _device_planning = True

    def __init__(self, learning_rate=1e-4, batch_size=64, max_epochs=500, patience=5, ncores=-1,
                 loss="wMSE", output_prefix=_SCRATCH, sub_outputdim=512, verbose=1, seed=1234,
                 architecture=None, device_id=0, comm=None, precision="fp32", stream_matrix=None):
        self.NN_parameters = dict(learning_rate=learning_rate, batch_size=batch_size, loss=loss,
                                  architecture=architecture, max_epochs=max_epochs, patience=patience)
        self.sub_outputdim = sub_outputdim
        self.outputdir = output_prefix
        self.verbose = verbose
        self.seed = seed
        self.device_id = device_id                 # extension: which GPU
        # extensions (BASELINE configs[4]): "bf16" stores the gathered predictor blocks in bfloat16, runs inference /
        # validation on the bf16 matrix cores and gives the training GEMMs bf16 operands wherever the engine has the
        # bf16 matrix-core kernel for them (engine.training_precision / engine.path_info() say where); weights (fp32
        # master copies), Adam state, targets and every accumulation stay fp32; stream_matrix
        # hands the log1p matrix over in row blocks through pinned buffers so that it never resides on the GPU
        # (None: automatically, for matrices above 32 GB)
        self.precision = precision
        self.stream_matrix = stream_matrix
        self._engine = None
        # extension: a deepimpute_amd.sharded Comm (one process per GPU); the string "rccl" builds
        # an RcclComm from RANK/WORLD_SIZE/LOCAL_RANK at fit time.  None = single process.
        self._comm_spec = comm                     # what the caller asked for
        self._comm = None                          # the live communicator of the current engine
        self._first_subnet = 0
        if isinstance(comm, str) and comm.lower() == "rccl":
            # one process per GPU: the rank's device is known before any planning touches a GPU
            self.device_id = int(os.environ.get("LOCAL_RANK", str(device_id)))
        self.setCores(ncores)

    def _warm_up(self):
        """The GPU comes up (HIP context, pinned bounce buffers, the finish pipeline's blocks: dimn_warm_up) on a helper thread while
        fit() / predict() still look at their frame.  Not in __init__: the reference's constructor has no side effects."""
        if self._device_planning and self._comm_spec is None:
            try:
                from . import _lib
                _lib.warm_up_async(self.device_id)
            except (ImportError, OSError):
                pass                                   # no library: the engine's constructor says so

    def setCores(self, ncores):
        # the reference only sizes TensorFlow's CPU thread pools with this (multinet.py:222-223);
        # kept as an attribute (and message) for compatibility, unused by the GPU path
        self.ncores = ncores if ncores > 0 else os.cpu_count()
        if ncores <= 0:
            print("Using all the cores ({})".format(self.ncores))

    def loadDefaultArchitecture(self):
        self.NN_parameters['architecture'] = [
            {"type": "dense", "neurons": self.sub_outputdim // 2, "activation": "relu"},
            {"type": "dropout", "rate": 0.2}]

    def _engine_classes(self):
        """(engine of the tuned kernels, engine of the general path): the two constructors build() chooses between.  The product
        binds libdimn.so (hand-written HIP) and has no other implementation behind this seam."""
        from .engine import HipEngine, HipGeneralEngine
        return HipEngine, HipGeneralEngine

    # -- the seam: where the reference builds/compiles the Keras model (multinet.py:126-167) --
    def build(self, inputdims, subnet_offset=0):
        if self.NN_parameters['architecture'] is None:
            self.loadDefaultArchitecture()
        print(self.NN_parameters['architecture'])
        layers = _parse_architecture(self.NN_parameters['architecture'])
        loss = _loss_name(self.NN_parameters['loss'])
        batch = int(self.NN_parameters["batch_size"])
        common = dict(batch_size=batch, learning_rate=self.NN_parameters["learning_rate"], seed=0 if self.seed is None else self.seed,
                      device_id=self.device_id, subnet_offset=subnet_offset)
        if str(self.precision).lower() not in ("fp32", "f32", "float32"):
            common["precision"] = self.precision
        tuned_cls, general_cls = self._engine_classes()
        # the tuned kernels take the reference's default shape family: one hidden layer of <= 384 units (+ dropout),
        # batch <= 64, wMSE -- loadDefaultArchitecture(), the CLI defaults; everything else build() accepts runs on the
        # general path (dimn_create_general)
        tuned = len(layers) == 1 and layers[0][0] <= 384 and batch <= 64 and loss in ("wmse", "wmse_binary")
        if tuned:
            hidden, act, rate = layers[0]
            extra = {} if act == "relu" else {"activation": act}
            if loss == "wmse_binary":
                extra["loss_binary"] = True
            return tuned_cls(list(inputdims), hidden, self.sub_outputdim, dropout_rate=rate, **common, **extra)
        if general_cls is None:
            raise NotImplementedError("no general engine for %r" % (self.NN_parameters['architecture'],))
        return general_cls(list(inputdims), layers, self.sub_outputdim, loss=loss, **common)

    # -- persistence (reference: model.json + model.h5, multinet.py:105-124) --
    def _model_format(self):
        """"h5" (the reference's pair: Keras model.json + Keras-layout model.h5) when the machine has an HDF5 library, else
        "npz" (model.json + model.npz).  DIMN_MODEL_FORMAT=npz|h5|both overrides."""
        from . import keras_io
        want = os.environ.get("DIMN_MODEL_FORMAT", "").lower()
        if want in ("h5", "both") and not keras_io.available():
            raise OSError("DIMN_MODEL_FORMAT=%s but no HDF5 library was found (set DIMN_LIBHDF5)" % want)
        return want if want in ("npz", "h5", "both") else ("h5" if keras_io.available() else "npz")

    def _sharded_outputdir(self, comm):
        """A sharded job writes per-rank shards into ONE directory.  The default output_prefix is a per-process temporary
        directory (the reference's `tempfile.mkdtemp()` default, multinet.py:60), which every rank would create for itself:
        with more than one rank it is replaced by a directory named after the job (launcher pid + start time + MASTER_PORT:
        the same string on every rank, sharded._job_tag) -- an explicit output_prefix is used as given."""
        if comm is not None and comm.world > 1 and self.outputdir == _SCRATCH:
            self.outputdir = self._job_directory(create=True)
        return self.outputdir

    @staticmethod
    def _job_directory(create):
        """The per-job default directory of a sharded fit: a predictable name under the shared temporary directory, so it is
        created private (0700) and refused unless it is a real directory (no symlink) owned by this user."""
        import stat
        from .sharded import _job_tag
        path = os.path.join(tempfile.gettempdir(), "dimn_model_%d_%s" % (os.getuid(), _job_tag()))
        if create:
            os.makedirs(path, mode=0o700, exist_ok=True)
        info = os.lstat(path)                        # (FileNotFoundError when a load() finds nothing: the caller's message)
        if not stat.S_ISDIR(info.st_mode) or stat.S_ISLNK(info.st_mode) or info.st_uid != os.getuid():
            raise PermissionError("MultiNet: %s is not a directory owned by uid %d; pass output_prefix explicitly" % (path, os.getuid()))
        return path

    def save(self, model):
        """model.json (rank 0; the Keras functional-model JSON of build()'s network, our own metadata under the extra key
        "deepimpute_amd") + the weights in Keras layout: model.h5 as Keras save_weights writes it (keras_io.py) or, without
        an HDF5 library, model.npz keyed by GLOBAL sub-net index.  A sharded job writes model.rank<r>.npz per rank (one
        node, one file system), from which rank 0 assembles model.h5; a fresh MultiNet under any world size can load() either."""
        from . import keras_io
        comm = self._comm
        rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
        self._sharded_outputdir(comm)
        _join_saves(self.outputdir)                      # (an earlier fit's files still being written into the same directory)
        os.makedirs(self.outputdir, exist_ok=True)
        fmt = self._model_format()
        layers = _parse_architecture(self.NN_parameters['architecture'])
        dims = [len(p) for p in self.predictors] if getattr(self, "predictors", None) is not None else list(model.D)
        if rank == 0:
            meta = {"format": "deepimpute_amd-3", "inputdims": dims, "sub_outputdim": self.sub_outputdim,
                    "architecture": self.NN_parameters['architecture'], "loss": _loss_name(self.NN_parameters['loss']),
                    "batch_size": int(self.NN_parameters["batch_size"]), "weights": fmt}
            with open(os.path.join(self.outputdir, "model.json"), "w") as fh:
                json.dump(keras_io.model_json(dims, layers, self.sub_outputdim, self.seed, meta), fh)
        blobs = {}
        import time
        laps = getattr(self, "timings", None)                # (fit.save split into its device and its file-system part: the latter varies 20x between boxes)
        t_fetch = time.perf_counter()
        # (the K device-to-host copies + re-layouts are independent reads of the handle: fetched on the host pool)
        fetched = _hostpar.pmap(model.get_weights, range(model.K)) if hasattr(model, "predict_device") else [model.get_weights(k) for k in range(model.K)]
        if isinstance(laps, dict):
            laps["fit.save.fetch_weights"] = laps.get("fit.save.fetch_weights", 0.0) + time.perf_counter() - t_fetch
        t_write = time.perf_counter()
        for k in range(model.K):                         # dense layer l (1-based, the last one is the output layer): Wl_<k>, bl_<k>
            arrays = fetched[k]
            for i, arr in enumerate(arrays):
                blobs["%s%d_%d" % ("Wb"[i % 2], i // 2 + 1, self._first_subnet + k)] = arr
        deferred = world == 1 and getattr(self, "_defer_save", False)
        if deferred:
            # fit(): the weights are on the host; writing them -- 119 MB of HDF5 at the 50k x 20k job, 0.03-0.05 s -- needs neither the GPU
            # nor the caller, so it runs on a helper thread UNDER fit()'s held-out metrics only.  fit() joins it before it returns
            # (_finish_deferred_save: the reference's save() is synchronous, multinet.py:105-115, :249 -- when fit() returns the files
            # exist, a failed write raises out of fit(), and "Saved model to disk" is printed after the write).  The helper times itself
            # into a private dict; the caller merges it into self.timings after the join.
            self._deferred_laps = {}
            _start_save(self.outputdir, lambda: self._write_weight_files(blobs, fmt, layers, dims, rank, world, comm, self._deferred_laps, t_write))
        else:
            self._write_weight_files(blobs, fmt, layers, dims, rank, world, comm, laps, t_write)
        written = {"h5": "model.json + model.h5 (Keras save_weights layout)", "npz": "model.json + model.npz (no HDF5 library found: set DIMN_LIBHDF5, "
                   "or DIMN_MODEL_FORMAT=h5 to insist)" if not os.environ.get("DIMN_MODEL_FORMAT") else "model.json + model.npz",
                   "both": "model.json + model.h5 + model.npz"}[fmt]
        message = "Saved model to disk in {}".format(self.outputdir) + " [%s%s]" % (written, "; shards model.rank0..%d.npz" % (world - 1) if world > 1 else "")
        if deferred:
            self._deferred_message = message
        else:
            print(message)

    def _finish_deferred_save(self):
        """fit()'s half of a deferred save(): wait for the weight files (a failed write raises HERE, out of fit()), merge the helper's
        lap times, print the reference's message (multinet.py:115) now that it is true."""
        try:
            _join_saves(self.outputdir)
        finally:
            laps, extra = getattr(self, "timings", None), self.__dict__.pop("_deferred_laps", None)
            if isinstance(laps, dict) and extra:
                for name, seconds in extra.items():
                    laps[name] = laps.get(name, 0.0) + seconds
        message = self.__dict__.pop("_deferred_message", None)
        if message:
            print(message)

    def _write_weight_files(self, blobs, fmt, layers, dims, rank, world, comm, laps, t_write):
        """The file-system half of save(): stale files out, model.npz / model.h5 (and the shards of a sharded job) in."""
        import time
        from . import keras_io
        # never leave weights of an older fit beside the new ones: the other format's file, and the shards of ranks this job
        # does not have (a refit into the same directory with fewer ranks would otherwise leave model.rank<r>.npz, r >= world,
        # whose sub-net indices overlap the fresh shards)
        if rank == 0:
            stale = [path for path in glob.glob(os.path.join(self.outputdir, "model.rank*.npz")) if _shard_rank(path) is None or _shard_rank(path) >= (world if world > 1 else 0)]
            if fmt != "both":
                stale.append(os.path.join(self.outputdir, "model.npz" if fmt == "h5" else "model.h5"))
            for path in stale:
                if os.path.exists(path):
                    os.remove(path)
        problem = ""
        if world > 1:
            np.savez(os.path.join(self.outputdir, "model.rank%d.npz" % rank), **blobs)
            comm.barrier()                               # every shard is on disk
            if rank == 0:
                blobs = {}
                for r in range(world):                   # exactly this job's shards, in rank order
                    path = os.path.join(self.outputdir, "model.rank%d.npz" % r)
                    if not os.path.exists(path):
                        problem += " %s is missing (ranks must share output_prefix=%r)" % (os.path.basename(path), self.outputdir)
                        continue
                    with np.load(path) as z:
                        blobs.update({f: z[f] for f in z.files})
                n_dense = len([l for l in layers if l[0] > 0]) + 1
                missing = [key for k in range(len(dims)) for l in range(1, n_dense + 1) for key in ("W%d_%d" % (l, k), "b%d_%d" % (l, k)) if key not in blobs]
                if missing and not problem:
                    problem = " the shards lack %d arrays (first: %s)" % (len(missing), missing[0])
            # every rank learns of a failed assembly BEFORE the final barrier, and every rank raises
            if comm.allreduce_sum(np.array([1.0 if problem else 0.0]))[0] > 0:
                raise OSError("MultiNet.save: rank 0 could not assemble the sharded weights in %s:%s" % (self.outputdir, problem or " (see rank 0)"))
        elif fmt != "h5":
            np.savez(os.path.join(self.outputdir, "model.npz"), **blobs)
        if rank == 0 and fmt != "npz":
            K = len(dims)
            inputs, hidden, drops, outputs = keras_io.layer_names(K, layers)
            order = list(inputs)
            for l in range(len(layers)):
                order += (hidden[l] or []) + (drops[l] or [])
            order += outputs
            weights = {}
            for k in range(K):
                for l, name in enumerate([h[k] for h in hidden if h is not None] + [outputs[k]], start=1):
                    weights[name] = (blobs["W%d_%d" % (l, k)], blobs["b%d_%d" % (l, k)])
            keras_io.write_weights_h5(os.path.join(self.outputdir, "model.h5"), order, weights)
        if comm is not None:
            comm.barrier()                               # every file is on disk when any rank returns
        if isinstance(laps, dict):
            laps["fit.save.write_files"] = laps.get("fit.save.write_files", 0.0) + time.perf_counter() - t_write

    def load(self):
        """The engine holding the fitted weights: the live one if this object trained it, else rebuilt from outputdir
        (weights only, like Keras load_weights: no optimizer state).  Reads what save() writes and the reference's own
        pair (a Keras model.json without our metadata + model.h5; loss and batch size then stay as constructed)."""
        if self._engine is None:
            from . import keras_io
            if isinstance(self._comm_spec, str) and self.outputdir == _SCRATCH and int(os.environ.get("WORLD_SIZE", "1")) > 1:
                self.outputdir = self._job_directory(create=False)      # the directory a sharded fit of this job wrote to (_sharded_outputdir)
            _join_saves(self.outputdir)                  # a fit() of this process may still be writing these files
            with open(os.path.join(self.outputdir, "model.json")) as fh:
                doc = json.load(fh)
            dense_names = None
            if "config" in doc:                          # a Keras functional-model JSON
                inputdims, arch, out_dim, dense_names = keras_io.parse_model_json(doc)
                meta = doc.get("deepimpute_amd") or {"inputdims": inputdims, "architecture": arch, "sub_outputdim": out_dim}
            else:
                meta = doc                               # format deepimpute_amd-2
            self.NN_parameters['architecture'] = meta["architecture"]
            self.sub_outputdim = meta["sub_outputdim"]
            if "loss" in meta:
                self.NN_parameters['loss'] = meta["loss"]
            if "batch_size" in meta:
                self.NN_parameters['batch_size'] = meta["batch_size"]
            engine, _, counts = self._build_shard(meta["inputdims"])
            wanted = set(range(self._first_subnet, self._first_subnet + engine.K))
            h5_path = os.path.join(self.outputdir, "model.h5")
            if dense_names is not None and os.path.exists(h5_path) and meta.get("weights", "h5") != "npz":
                by_layer = keras_io.read_weights_h5(h5_path, only={name for g in wanted for name in dense_names[g]})
                for g in sorted(wanted):
                    arrays = [a for name in dense_names[g] for a in by_layer[name]]
                    engine.set_weights(g - self._first_subnet, *arrays)
                wanted.clear()
            else:
                shards = [path for path in glob.glob(os.path.join(self.outputdir, "model.rank*.npz")) if _shard_rank(path) is not None]
                files = sorted(shards, key=_shard_rank) or [os.path.join(self.outputdir, "model.npz")]
                for path in files:
                    with np.load(path) as z:
                        for g in sorted(wanted):
                            if "W1_%d" % g in z.files:
                                n_dense = sum(1 for f in z.files if f.startswith("W") and f.endswith("_%d" % g))
                                arrays = [z["%s%d_%d" % (wb, l, g)] for l in range(1, n_dense + 1) for wb in "Wb"]
                                engine.set_weights(g - self._first_subnet, *arrays)
                                wanted.discard(g)
            if wanted:
                raise FileNotFoundError("weights of sub-networks %s not found in %s" % (sorted(wanted), self.outputdir))
            self._engine = engine
            self._counts = counts
        return self._engine

    def _hand_over(self, engine, norm, with_targets):
        """The log1p matrix to the engine + the device gather of every sub-net's blocks; streamed in row blocks when asked
        for (or automatically above 32 GB) on engines that can."""
        stream = self.stream_matrix
        if stream is None:
            stream = norm.size * 4 > (32 << 30)
        if stream and hasattr(engine, "predict_device"):          # HIP engines
            engine.set_matrix(norm, streamed=True, with_targets=with_targets)
        else:
            engine.set_matrix(norm)
            engine.gather(with_targets)

    def _bind_columns(self, engine, columns):
        """Translate gene labels of predictors/targets into column positions of the matrix."""
        where = pd.Index(columns)
        for k in range(engine.K):
            g = self._first_subnet + k                   # engine-local -> global sub-net
            cols_in, cols_out = where.get_indexer(self.predictors[g]), where.get_indexer(self.targets[g])
            if (cols_in < 0).any() or (cols_out < 0).any():
                raise KeyError("predictor/target genes missing from the data columns")
            engine.set_indices(k, cols_in, cols_out)

    # -- fit: planning on the host, training on the GPU --
    def fit(self, raw, cell_subset=1, NN_lim=None, genes_to_impute=None, n_pred=None, ntop=5,
            minVMR=0.5, mode='random'):
        tm = self.timings = _Stages()
        self._warm_up()
        with tm.stage("fit.as_float64"):
            raw = self._as_count_frame(raw)
        # Fast path (a frame of raw counts, one GPU, resident matrix): the counts go to the device first, as float32 (exact), and
        # everything the planning needs is computed from that copy -- the gene statistics (pandas' additions in pandas' order, one
        # thread per gene: DataFrame.mean() / .var() to the bit), the candidate pool of the correlation (genes that vary, with a
        # positive mean) and, on a helper thread while this one picks the genes, the g x g correlation itself (exact integer
        # arithmetic on the int8 matrix cores).  Every number is the one the plain sequence below computes (checked where a guess
        # is involved).  [Round 2 ran the upload BESIDE host statistics: the two compete for host memory bandwidth, 0.45 s
        # together against 0.17 + 0.18 s one after the other; tools/dropin_probe.py.]
        upload, spec_pool, first, dev_early = None, None, None, None
        if cell_subset == 1 and self._counts_path_applies(raw):
            self._drop_resident()
            with tm.stage("fit.counts_upload"):
                from ._counts import DeviceCounts
                dev_early = DeviceCounts.try_create(raw.values, self.device_id)
            if dev_early is not None:
                with tm.stage("fit.gene_statistics"):
                    if raw.shape[0] < 2 or not _hostpar.pandas_order_holds():
                        first = None                         # (pandas' var of one row / another pandas: the plain sequence below computes what that gives)
                    elif os.environ.get("DIMN_DEVICE_STATS", "1") != "0" or raw.values.dtype != np.float64:
                        first = dev_early.gene_stats()
                    else:                                    # the host routines (same numbers; dimn_hoststats.h)
                        first = _hostpar.col_stats_first(raw.values)
                        if first is not None:
                            first["var"] = _hostpar.col_stats_var(raw.values, first["avg"])
        if first is not None:
            with tm.stage("fit.inspect_data"):           # (before any helper thread has work in flight: inspect_data() may exit(1))
                inspect_data(raw, _max=first["vmax"])
            if n_pred is None and ntop <= 16:            # (the device selection takes ntop <= 16: nothing is computed ahead that it would not use)
                spec_pool = np.flatnonzero((first["cmax"] > first["cmin"]) & (first["mean"] > 0)).astype(np.int32)
            upload = self._start_correlation(dev_early, spec_pool, raw.shape[0])
            if self.seed is not None:
                np.random.seed(self.seed)
            var = pd.Series(first["var"], index=raw.columns)
            mean = pd.Series(first["mean"], index=raw.columns)
        else:
            upload = (lambda: dev_early)                     # (None unless the statistics declined after a successful upload)
            with tm.stage("fit.inspect_data"):
                inspect_data(raw)
            if self.seed is not None:
                np.random.seed(self.seed)
            if cell_subset != 1:
                # fraction below 1, absolute cell count otherwise (the CLI passes a float)
                raw = raw.sample(frac=cell_subset) if cell_subset < 1 else raw.sample(int(cell_subset))

            # variance over (1 + mean), most variable first, strictly positive (multinet.py:191-192)
            with tm.stage("fit.gene_statistics"):
                var, mean = _hostpar.column_var_mean(raw)
        with tm.stage("fit.gene_selection"):
            gene_metric = (var / (1 + mean)).sort_values(ascending=False)
            gene_metric = gene_metric[gene_metric > 0]
            if genes_to_impute is None:
                genes_to_impute = self.filter_genes(gene_metric, minVMR, NN_lim=NN_lim)
            else:
                genes_to_impute = self._pad_gene_list(genes_to_impute, gene_metric)

        # setTargets only looks at the column labels; the reference hands it raw.reindex(columns=...),
        # a full copy of the matrix (multinet.py:212) -- `_ColumnsOnly` carries the same labels (even an empty
        # DataFrame with 20k columns costs pandas half a second to build)
        with tm.stage("fit.gene_selection"):
            self.setTargets(_ColumnsOnly(genes_to_impute), mode=mode)
        # get_distance_matrix + setPredictors (multinet.py:211-214; neither draws random numbers, so their order
        # against setTargets is free): fused on the GPU -- the g x g correlation never comes back to the host --
        # whenever a GPU is visible; otherwise, and for the shapes the kernel does not take, the two public
        # functions below run as in the reference.
        with tm.stage("fit.correlation_wait"):
            dev_counts = upload()
        with tm.stage("fit.correlation+predictors"):
            on_device = _gpu_visible() and self._set_predictors_device(raw, n_pred, ntop, (var, mean), counts=dev_counts, corr_pool=spec_pool)
            if dev_counts is not None:
                dev_counts.corr_drop()                   # whatever the helper thread left that the selection did not consume (pool^2 * 8 bytes)
            if not on_device:
                correlations = get_distance_matrix(raw, n_pred=n_pred, device_id=self.device_id, _var_mean=(var, mean))
                self.setPredictors(correlations, ntop=ntop)

        print("Normalization")
        # A sharded job whose frame is the node's ONE shared copy (_shm.share_frame; DIMN_SHARE_NORM=1 forces it for private frames too) makes
        # the log1p matrix ONCE as well: each rank fills its slice of the rows of a /dev/shm segment -- that needs the communicator, which is
        # bound to the engine, so it happens right after "Building network" (log1p draws no random numbers: the order is free)
        from . import _shm
        share_norm = (dev_counts is None and self._comm_spec is not None
                      and (_shm.shared_of(raw.values) is not None or os.environ.get("DIMN_SHARE_NORM") == "1"))
        with tm.stage("fit.log1p"):
            # with the counts resident, log1p happens on the device (numpy's table); the frame below then only carries the labels
            norm_data = None if share_norm else (_hostpar.log1p_float32(raw) if dev_counts is None else _ColumnsOnly(raw.columns, raw.index, raw.shape))
        np.random.seed(self.seed)                      # second seeding, multinet.py:219

        print("Building network")
        with tm.stage("fit.build"):
            self._release_engine()
            engine, comm, counts = self._build_shard([len(p) for p in self.predictors])
        if norm_data is None:
            with tm.stage("fit.log1p"):
                if comm.world > 1:
                    norm_data = pd.DataFrame(_shm.shared_log1p(raw, comm), index=raw.index, columns=raw.columns, copy=False)
                else:
                    norm_data = _hostpar.log1p_float32(raw)
        if comm.world > 1 and hasattr(engine, "set_stream_order"):
            engine.set_stream_order(comm.rank, comm.world)       # the ranks' streamed hand-overs walk different pages of the shared matrix
        t_split = tm.stage("fit.split")
        t_split.__enter__()

        held_out = np.random.choice(norm_data.index, int(_VALIDATION_FRACTION * norm_data.shape[0]), replace=False)
        # train_cells = np.setdiff1d(index, test_cells) (multinet.py:229) is label-sorted and unique; inspect_data()
        # has made sure the labels are unique, so the same rows come out of one argsort of the labels
        # (np.setdiff1d on 50k string labels takes 0.7 s)
        rows_val = norm_data.index.get_indexer(held_out)
        is_train = np.ones(norm_data.shape[0], bool)
        is_train[rows_val] = False
        by_label = np.argsort(norm_data.index.values, kind="stable")
        rows_train = by_label[is_train[by_label]]
        t_split.__exit__(None, None, None)

        with tm.stage("fit.hand_over"):
            self._bind_columns(engine, norm_data.columns)
            if dev_counts is not None and hasattr(engine, "set_matrix_counts"):
                engine.set_matrix_counts(dev_counts)
                engine.gather(True)
                self._resident = (dev_counts, raw.columns)   # predict() of the same frame finds everything in place
            else:
                if dev_counts is not None:                   # an engine without the entry point (tests: oracle engines)
                    dev_counts.close()
                    dev_counts, norm_data = None, _hostpar.log1p_float32(raw)
                self._hand_over(engine, norm_data.values, True)
            engine.set_split(rows_train, rows_val)
            engine.init_weights(0 if self.seed is None else self.seed)

        print("Fitting with {} cells".format(norm_data.shape[0]))
        plan_thread = self._start_predict_plan(raw.columns)      # (np.unique over the K * O target labels: host work that fits under the training)
        with tm.stage("fit.train"):
            if comm.world == 1:
                epochs, loss_curve, val_curve = engine.fit(self.NN_parameters["max_epochs"], self.NN_parameters["patience"])
            else:
                from .sharded import fit_sharded
                epochs, loss_curve, val_curve = fit_sharded(engine, comm, self.NN_parameters["max_epochs"],
                                                            self.NN_parameters["patience"])
        self.history = {"loss": [float(x) for x in loss_curve], "val_loss": [float(x) for x in val_curve]}
        if self.verbose:
            for i, (a, b) in enumerate(zip(loss_curve, val_curve), start=1):
                print("Epoch {}/{} - loss: {:.4f} - val_loss: {:.4f}".format(i, self.NN_parameters["max_epochs"], a, b))
        self.trained_epochs = int(epochs)
        print("Stopped fitting after {} epochs".format(self.trained_epochs))
        plan_thread.join()

        self._engine = engine
        self._counts = counts
        with tm.stage("fit.save"):
            self._defer_save = True                      # the files are written behind fit()'s back (save() called directly writes them before it returns)
            try:
                self.save(engine)
            finally:
                self._defer_save = False
        try:
            with tm.stage("fit.held_out_metrics"):
                self.test_metrics = self._held_out_metrics(engine, norm_data, held_out, rows_val)
        finally:
            with tm.stage("fit.save.join"):
                self._finish_deferred_save()             # the files are on disk (or the write's error is raised) before fit() returns
        with tm.stage("fit.free"):
            if share_norm and hasattr(norm_data, "values"):
                _shm.release(norm_data.values)               # (the node's shared log1p segment: unmapped here, freed when the last rank has done so)
            del norm_data, var, mean, gene_metric
        return self

    def _predict_plan(self, columns):
        """(genes, slot_gene, where) of predict(): a gene may occupy several target slots -- the unique genes (label-sorted, like the
        reference's groupby(columns).mean(), multinet.py:282-284), the gene of every slot, and the genes' columns in the frame."""
        slots = self.targets.flatten()
        genes, slot_gene = np.unique(slots, return_inverse=True)
        where = pd.Index(columns).get_indexer(genes)
        return genes, slot_gene, where

    def _start_predict_plan(self, columns):
        """Compute predict()'s plan for a frame with fit()'s columns on a helper thread while the GPU trains; predict() takes it when its
        frame has those columns and the targets are still the ones planned for."""
        import threading
        self._plan_cache = None
        targets = np.array(self.targets, copy=True)

        def work():
            try:
                self._plan_cache = (columns, targets, self._predict_plan(columns))
            except Exception:
                self._plan_cache = None                  # predict() computes it itself
        thread = threading.Thread(target=work, name="dimn-predict-plan", daemon=True)
        thread.start()
        return thread

    def _as_count_frame(self, raw):
        """An INTEGER frame -- what pd.read_csv / the CLI's reader make of a count matrix (deepImpute.py:13) -- takes the resident-counts
        path like a float64 frame does: one upload, statistics / correlation / log1p / restore from the device copy.  A C-ordered int64
        frame (the CLI's reader) is read in place; any other integer layout (pandas' own reader builds column-major blocks) becomes the
        float64, C-ordered frame of the same numbers first (by row blocks on the host pool).  The reference's own
        arithmetic converts the same way (np.log1p(raw), raw.var(): float64); positive counts come back from predict() as float64, as
        multinet.py:296-303 returns them.  Anything else (float frames, object columns, a sharded or streamed job) passes through."""
        values = getattr(raw, "values", None)
        if (not isinstance(raw, pd.DataFrame) or not isinstance(values, np.ndarray) or values.ndim != 2 or values.dtype.kind not in "iu"
                or (values.dtype == np.int64 and values.flags.c_contiguous)      # read in place (dimn_counts_create_typed): no copy at all
                or os.environ.get("DIMN_RESIDENT_COUNTS", "1") == "0" or self._comm_spec is not None or not self._device_planning
                or self.stream_matrix or values.size * 4 > (32 << 30) or not _gpu_visible()):
            return raw
        return pd.DataFrame(_hostpar.as_float64(values), index=raw.index, columns=raw.columns, copy=False)

    def _counts_path_applies(self, raw):
        """The resident-counts path is for: one GPU (no sharded / streamed job), the product engine, a C-ordered float64 frame
        that fits the device (DIMN_RESIDENT_COUNTS=0 switches it off).  Whether the VALUES are counts is decided by the upload."""
        values = getattr(raw, "values", None)
        from ._cabi import count_dtype
        return not (os.environ.get("DIMN_RESIDENT_COUNTS", "1") == "0" or self._comm_spec is not None or not self._device_planning or self.stream_matrix
                    or count_dtype(values) is None or values.size * 4 > (32 << 30) or not _gpu_visible())

    def _start_correlation(self, dev, pool, n_cells):
        """|corr| of the candidate pool on a helper thread (ctypes releases the GIL; the work is on the GPU) while the caller
        ranks genes on the host; returns a function that waits for it and gives `dev` back.  pool None: nothing to run ahead."""
        import threading
        box = {}

        def work():
            try:
                if pool is not None and 2 <= len(pool) <= 65535 and n_cells >= 2:
                    dev.corr(pool)
                    dev.corr_ready = True
            except Exception as exc:                          # the selection then runs the product itself and reports a real failure
                box["error"] = exc
        thread = threading.Thread(target=work, name="dimn-correlation", daemon=True)
        thread.start()

        def wait():
            thread.join()
            if "error" in box:                               # not fatal: the selection runs the product itself and raises a real failure there
                warnings.warn("deepimpute_amd: the correlation computed ahead of the gene selection failed (%r); recomputing" % (box["error"],), RuntimeWarning)
            return dev
        return wait

    def _start_counts_upload(self, raw):
        """Begin uploading raw's counts to the GPU on a helper thread (ctypes releases the GIL) and return a function that waits
        for it and gives the _counts.DeviceCounts, or None where the fast path does not apply / the values are not counts
        (predict() of a frame that is not the fitted one: the upload runs beside load())."""
        self._drop_resident()
        values = getattr(raw, "values", None)
        if not self._counts_path_applies(raw):
            return lambda: None
        import threading
        from ._counts import DeviceCounts
        box = {}

        def work():
            try:
                box["counts"] = DeviceCounts.try_create(values, self.device_id)
            except Exception as exc:                          # the host path takes over; a real failure shows up there
                box["error"] = exc
        thread = threading.Thread(target=work, name="dimn-counts-upload", daemon=True)
        thread.start()

        def wait():
            thread.join()
            return box.get("counts")
        return wait

    def _start_checksum(self, counts, values):
        """counts.matches(values) on a helper thread (ctypes releases the GIL); returns a function that waits for the answer."""
        import threading
        box = {}

        def work():
            try:
                box["same"] = counts.matches(values)
            except Exception:
                box["same"] = False
        thread = threading.Thread(target=work, name="dimn-checksum", daemon=True)
        thread.start()

        def wait():
            thread.join()
            return bool(box.get("same"))
        return wait

    def _drop_resident(self):
        held = getattr(self, "_resident", None)
        self._resident = None
        if held is not None:
            held[0].close()

    def _release_engine(self):
        """Close the live communicator (a collective: every rank of a sharded job calls fit/close alike),
        then the engine it was bound to."""
        if self._comm is not None:
            self._comm.close()
            self._comm = None
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        self._drop_resident()                           # (after the engine that read them)

    def close(self, release_cache=True):
        """Extension: release the GPU (and, in a sharded job, the RCCL communicator) now instead of at exit -- the engine, the resident
        counts, and the library's process-wide cache of large device blocks (which otherwise waits for the next fit() of this process:
        deepimpute_amd.release_cached_memory; release_cache=False leaves it for a fit() that follows)."""
        self._release_engine()
        _join_saves(self.outputdir)
        if not release_cache:
            return
        try:
            from . import _lib
            _lib.release_cached_memory()
        except (ImportError, OSError):
            pass

    def _shard_plan(self, inputdims):
        """(rank, world, counts) of this process for the sub-networks of `inputdims` under the caller's comm spec (contiguous blocks
        balanced by predictor count D_k: sharded.shard_subnets)."""
        K = len(inputdims)
        from .sharded import shard_subnets
        spec = self._comm_spec
        if spec is None:
            rank, world = 0, 1
        elif isinstance(spec, str):
            if spec.lower() != "rccl":
                raise ValueError("comm must be a Comm object or 'rccl'")
            rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        else:
            rank, world = spec.rank, spec.world
        counts, _ = shard_subnets(K, world, weights=inputdims if world > 1 and K > world else None)
        if min(counts) < 1:
            raise ValueError("more ranks (%d) than sub-networks (%d)" % (world, K))
        return rank, world, counts

    def _build_shard(self, inputdims):
        """The engine of this rank's contiguous block of sub-nets + the communicator bound to it."""
        from .sharded import RcclComm, SingleComm
        rank, world, counts = self._shard_plan(list(inputdims))
        self._first_subnet = sum(counts[:rank])
        mine = range(self._first_subnet, self._first_subnet + counts[rank])
        engine = self.build([inputdims[k] for k in mine], subnet_offset=self._first_subnet)
        spec = self._comm_spec
        if spec is None:
            comm = SingleComm()
        elif isinstance(spec, str):
            comm = RcclComm(engine, rank, world)       # bound to THIS engine's handle; closed with it
        else:
            comm = spec
        self._comm = comm
        return engine, comm, counts

    def _predict_block(self, engine, rows=None):
        """np.hstack of ALL sub-nets' outputs on rank 0 (None elsewhere when sharded)."""
        comm = self._comm
        if comm is None or comm.world == 1:
            return engine.predict(rows)
        from .sharded import predict_sharded
        return predict_sharded(engine, comm, self._counts, rows)

    def _pad_gene_list(self, genes, gene_metric):
        """User-supplied gene list made a multiple of sub_outputdim (multinet.py:196-209)."""
        count = len(genes)
        if count % self.sub_outputdim == 0:
            return genes
        print("The number of input genes is not a multiple of {}. Filling with other genes.".format(count))
        wanted = self.sub_outputdim - count
        extra = gene_metric.index[:wanted]
        if len(extra) < wanted:
            extra = np.concatenate([extra, np.random.choice(gene_metric.index, wanted - len(extra), replace=True)])
        return np.concatenate([genes, extra])

    def _held_out_metrics(self, engine, norm_data, held_out, rows_val):
        """Pearson r and MSE on the positive entries of the validation targets (multinet.py:251-262)."""
        comm = self._comm
        if hasattr(engine, "val_metrics") and (comm is None or comm.world == 1 or hasattr(comm, "allreduce_sum")):
            # the seven sums on the device (float64 accumulation over the resident predictions and targets; the
            # reference's scipy.stats.pearsonr runs in float32 over the same ~n_val*K*O values), summed over the ranks
            m = engine.val_metrics()
            if comm is not None and comm.world > 1:
                m = comm.allreduce_sum(m)
                if comm.rank != 0:
                    return None
            cnt, sx, sy, sxx, syy, sxy, sse = (float(v) for v in m)
            cov, vx, vy = sxy - sx * sy / cnt, sxx - sx * sx / cnt, syy - sy * sy / cnt
            return {'correlation': np.float32(cov / np.sqrt(vx * vy)), 'MSE': np.float32(sse / cnt)}
        guess = self._predict_block(engine, rows_val)
        if guess is None:                     # sharded job: only rank 0 holds the gathered block
            return None
        table, where = norm_data.values, norm_data.columns
        truth = np.hstack([table[np.ix_(rows_val, where.get_indexer(genes))] for genes in self.targets]).flatten()
        guess = guess.flatten()
        positive = truth > 0
        truth, guess = truth[positive], guess[positive]
        from scipy.stats import pearsonr                 # (imported where it is used: scipy.stats is a third of the package's import time, and
        #                                                   the product path computes these sums on the device)
        return {'correlation': pearsonr(truth, guess)[0],
                'MSE': np.sum((truth - guess) ** 2) / len(truth)}

    # -- predict: forward on the GPU, post-processing as multinet.py:282-310 --
    def predict(self, raw, imputed_only=False, policy="restore"):
        tm = self.timings = _Stages(getattr(self, "timings", None) or {})
        for key in [k for k in tm if k.startswith("predict.")]:
            del tm[key]
        self._warm_up()
        with tm.stage("predict.as_float64"):
            raw = self._as_count_frame(raw)
        with tm.stage("predict.load"):
            engine = self.load()
        # The counts of this very frame may still be on the GPU from fit() (verified bit for bit by a checksum pass), or go there
        # now in one upload; the engine then takes log1p, the forward pass and the restore / max step from that one copy.
        resident = None
        with tm.stage("predict.counts"):
            held = getattr(self, "_resident", None)
            values = getattr(raw, "values", None)
            verdict = None
            if held is not None and getattr(engine, "_dev_counts", None) is held[0] and raw.columns.equals(held[1]) and held[0].shape_matches(values):
                # same cells and columns; whether they are the same NUMBERS is one pass over the frame (a checksum of its bit patterns),
                # which runs on a helper thread beside the forward pass and the epilogue: a frame that turns out to differ costs a second,
                # ordinary predict below -- the result of the speculative one is dropped
                resident = held[0]
                # (policy "restore" on the device epilogue reads every element of the frame anyway -- dimn_impute_finish_restore -- and
                #  returns that checksum itself: no second pass over the 8 GB)
                folded = policy == "restore" and getattr(engine, "restore_epilogue", False)
                verdict = None if folded else self._start_checksum(resident, values)
            else:
                wait = self._start_counts_upload(raw) if hasattr(engine, "set_matrix_counts") else (lambda: None)
                fresh = wait()
                if fresh is not None:
                    self._bind_columns(engine, raw.columns)
                    engine.set_matrix_counts(fresh)
                    engine.gather(False)
                    self._resident = (fresh, raw.columns)
                    resident = fresh
        if resident is None:
            self._bind_columns(engine, raw.columns)
            with tm.stage("predict.log1p"):
                from . import _shm
                comm = self._comm
                if comm is not None and comm.world > 1 and (_shm.shared_of(raw.values) is not None or os.environ.get("DIMN_SHARE_NORM") == "1"):
                    norm = _shm.shared_log1p(raw, comm)                            # one copy for the node's ranks (as in fit())
                    if hasattr(engine, "set_stream_order"):
                        engine.set_stream_order(comm.rank, comm.world)
                else:
                    norm = _hostpar.log1p_float32(raw).values                      # float32(log1p(raw)): what Keras is fed
            with tm.stage("predict.hand_over"):
                self._hand_over(engine, norm, False)
            with tm.stage("predict.free"):
                from . import _shm as _shm_mod
                _shm_mod.release(norm)                                             # (a shared segment: forget it; a private array: no-op)
                del norm                                                           # (4 GB at 50k x 20k: unmapping it is not free)
        t_plan = tm.stage("predict.plan")
        t_plan.__enter__()
        # a gene may occupy several target slots: average them; the averaged columns are label-sorted,
        # like the reference's groupby(columns).mean() (multinet.py:282-284)
        cached = getattr(self, "_plan_cache", None)
        if cached is not None and (raw.columns is cached[0] or raw.columns.equals(cached[0])) and np.array_equal(np.asarray(self.targets), cached[1]):
            genes, slot_gene, where = cached[2]
        else:
            genes, slot_gene, where = self._predict_plan(raw.columns)
        if policy == "restore":
            print("Filling zeros")
        elif policy == "max":
            print("Imputing data with 'max' policy")
        observed = raw.values
        t_plan.__exit__(None, None, None)
        with tm.stage("predict.matrix_max"):
            top = resident.vmax if resident is not None else _hostpar.matrix_max(observed)
            ceiling = 2 * np.log1p(top)                               # overflow guard, multinet.py:292 (log1p is monotonic)

        with tm.stage("predict.forward+finish"):
            engine.last_observed_checksum = None         # (set by the restore epilogue when it has read the whole frame)
            from .engine import FrameMismatch
            same = True
            try:
                values = self._finish_on_device(engine, observed, where[slot_gene], policy, ceiling, resident=resident is not None)
            except FrameMismatch:                        # the restore epilogue found the frame to differ from the resident counts: straight to the re-upload
                values, same = None, False
            if same and values is not None and resident is not None and held is not None and resident is held[0]:
                same = verdict() if verdict is not None else getattr(engine, "last_observed_checksum", None) == resident.checksum
            if not same:
                # the frame is not the one that was fitted: upload it and run the ordinary sequence
                values = None
                fresh = self._start_counts_upload(raw)()
                if fresh is not None:
                    self._bind_columns(engine, raw.columns)
                    engine.set_matrix_counts(fresh)
                    engine.gather(False)
                    self._resident = (fresh, raw.columns)
                    top = fresh.vmax
                else:
                    self._bind_columns(engine, raw.columns)
                    self._hand_over(engine, _hostpar.log1p_float32(raw).values, False)
                    top = _hostpar.matrix_max(observed)
                ceiling = 2 * np.log1p(top)
                values = self._finish_on_device(engine, observed, where[slot_gene], policy, ceiling, resident=fresh is not None)
        if values is False:
            return None                                  # sharded job: rank 0 returns the frame
        if values is None:                               # engines without the device epilogue (tests: oracle / fake engines)
            block = self._predict_block(engine)          # [cells, K*O], np.hstack of the K outputs
            if block is None:
                return None
            values = self._finish_on_host(block, observed, slot_gene, len(genes), where, policy, ceiling)
        with tm.stage("predict.frame"):
            imputed = pd.DataFrame(values, index=raw.index, columns=raw.columns)
            return imputed.loc[:, genes] if imputed_only else imputed

    def _finish_on_device(self, engine, observed, slot_col, policy, ceiling, resident=False):
        """predict()'s post-processing as the device epilogue dimn_impute_finish (multinet.py:282-305): the K*O network
        outputs never come to the host, only the finished [cells, genes] float64 frame does.  None: this engine has no
        such epilogue (the host path runs); False: sharded job, this rank is not the root."""
        if not hasattr(engine, "impute_finish") or policy not in (None, "restore", "max") or (slot_col < 0).any():
            return None
        order = np.lexsort((np.arange(len(slot_col)), slot_col))              # slots grouped by output column, ascending slot order
        gene_off = np.zeros(observed.shape[1] + 1, np.int64)
        np.cumsum(np.bincount(slot_col, minlength=observed.shape[1]), out=gene_off[1:])
        comm = self._comm
        sharded = comm is not None and comm.world > 1
        if sharded and not getattr(comm, "device_gather", False):
            return None
        engine.predict_device()
        if sharded:
            engine.comm_gather_predictions(engine.n_cells, self._counts, root=0, is_root=False)   # stays in root's HBM
            if comm.rank != 0:
                return False
        if resident:                                     # the observed counts are the engine's resident matrix: nothing to upload
            return engine.impute_finish(None, gene_off, order, policy, ceiling, from_gathered=sharded, observed=observed)
        return engine.impute_finish(_hostpar.as_float64(observed), gene_off, order, policy, ceiling, from_gathered=sharded)

    def _finish_on_host(self, block, observed, slot_gene, n_genes, where, policy, ceiling):
        """The same post-processing with numpy on row blocks from the host pool (engines without the device epilogue)."""
        per_gene = np.bincount(slot_gene, minlength=n_genes).astype(np.float32)
        n_slots = len(slot_gene)
        first_slot = np.full(n_genes, -1, np.int64)
        first_slot[slot_gene[::-1]] = np.arange(n_slots - 1, -1, -1)          # lowest slot of each gene
        later = np.flatnonzero(first_slot[slot_gene] != np.arange(n_slots))   # the repeats, ascending
        # the reference concatenates predicted and untouched genes and re-orders them to raw's layout
        # (multinet.py:285-289); writing the averaged columns into a copy of log1p(raw) is the same
        # matrix.  Every step below is per cell, so it runs on row blocks from the host pool.
        values = np.empty(observed.shape, dtype=np.float64)
        untouched = len(where) < observed.shape[1]       # genes no sub-network predicts keep log1p(raw)

        def finish(ab):
            part = block[ab[0]:ab[1]]
            acc = part[:, first_slot]                    # float32 sums in slot order, as np.add.at would
            for s in later:
                acc[:, slot_gene[s]] += part[:, s]
            acc /= per_gene
            v = values[ab[0]:ab[1]]
            if untouched:
                v[...] = np.log1p(observed[ab[0]:ab[1]])
            v[:, where] = acc
            v[(v > ceiling) | np.isnan(v)] = 0
            np.expm1(v, out=v)                           # back to counts
            seen = observed[ab[0]:ab[1]]
            if policy == "restore":
                keep_raw = seen > 0
                v[keep_raw] = seen[keep_raw]
            elif policy == "max":
                keep_raw = seen > v
                v[keep_raw] = seen[keep_raw]
        _hostpar.pmap(finish, _hostpar.spans(len(values), _POST_ROWS))
        return values

    # -- planning helpers (public in the reference, so public here) --
    def filter_genes(self, gene_metric, threshold, NN_lim=None):
        """Genes to impute: the NN_lim most variable ones (default: all above `threshold`),
        rounded up to whole sub-networks and topped up with random genes (multinet.py:312-331;
        note the top-up is a full extra sub-network when the count is already a multiple)."""
        limit = int(NN_lim) if str(NN_lim).isdigit() else int((gene_metric > threshold).sum())
        width = self.sub_outputdim
        chosen = gene_metric.index[:int(np.ceil(limit / width)) * width]
        missing = width - (len(chosen) % width)
        if missing > 0:
            chosen = np.concatenate([chosen, np.random.choice(gene_metric.index, missing)])
        print("{} genes selected for imputation".format(len(chosen)))
        return chosen

    def setTargets(self, data, mode='random'):
        """Partition the genes to impute into K rows of sub_outputdim targets (multinet.py:333-342)."""
        shape = [int(data.shape[1] / self.sub_outputdim), self.sub_outputdim]
        if mode == 'progressive':
            self.targets = data.columns.values.reshape(shape)
        else:
            self.targets = np.random.choice(data.columns, shape, replace=False)

    def _set_predictors_device(self, raw, n_pred, ntop, var_mean, counts=None, corr_pool=None):
        """get_distance_matrix + setPredictors as ONE device job (dimn_select_predictors: fp64-MFMA |corr| of the
        candidate genes, then a top-`ntop` select per target over the resident matrix; multinet.py:20-34, 344-365).
        Same predictor lists as setPredictors() -- |corr| descending, ties in label order, first-occurrence unique --
        without the g x g host copy.  Returns False (nothing done) for what the kernel does not take: ntop > 16,
        repeated pool labels, targets outside the pool (the host path raises the reference's KeyError), a sub-net
        whose targets cover the whole pool (the reference's warning path)."""
        if ntop > 16:
            return False
        pool, values = _candidate_pool(raw, n_pred, var_mean, labels_only=counts is not None)
        if not pool.is_unique or raw.shape[0] < 2 or len(pool) > 65535:      # (one grid row per candidate gene in the finishing kernels)
            return False
        targets = np.asarray(self.targets)
        K, O = targets.shape
        rows = pool.get_indexer(targets.reshape(-1))
        if (rows < 0).any():
            return False
        rows = rows.reshape(K, O).astype(np.int32)
        if any(len(np.unique(r)) >= len(pool) for r in rows):
            return False
        from . import _cabi, _lib
        fns = _lib.load()
        rank = np.empty(len(pool), np.int32)
        rank[np.argsort(pool.values, kind="stable")] = np.arange(len(pool), dtype=np.int32)
        if counts is not None:                       # the candidate columns are read from the resident counts: no 8 GB upload
            pool_cols = raw.columns.get_indexer(pool).astype(np.int32)
            if corr_pool is not None and getattr(counts, "corr_ready", False) and np.array_equal(pool_cols, corr_pool):
                picks = counts.topk(rows, rank, ntop)          # the matrix product ran beside the gene statistics, over exactly this pool
            else:
                picks = counts.select_predictors(pool_cols, rows, rank, ntop)
        else:
            x = _hostpar.as_float64(values)
            picks = np.empty((K, O, ntop), np.int32)
            rc = fns["select_predictors"](int(self.device_id), _cabi.p_f64(x), x.shape[0], x.shape[1], _cabi.p_i32(rows), K, O,
                                          _cabi.p_i32(rank), int(ntop), _cabi.p_i32(picks))
            if rc != 0:
                raise RuntimeError("dimn_select_predictors: " + fns["last_error"]().decode("utf-8", "replace"))
        self.predictors = []
        for net in range(K):
            flat = picks[net].reshape(-1)
            chosen = pool[pd.unique(flat[flat >= 0])]                  # first-occurrence order, multinet.py:362
            self.predictors.append(chosen)
            print("Net {}: {} predictors, {} targets".format(net, len(chosen), len(self.targets[net])))
        return True

    def setPredictors(self, covariance_matrix, ntop=5):
        """Per sub-network: for each target the `ntop` most correlated genes outside the target
        set, in first-occurrence order (multinet.py:344-365).  The reference argsorts every full
        row; picking the top-ntop by partial selection yields the same genes in the same order
        whenever the correlations involved are distinct."""
        pool = covariance_matrix.columns
        table = covariance_matrix.values               # g x g float64; indexed by position below (pandas .loc
                                                       # on a 20k x 20k frame costs seconds per sub-network)
        by_label = np.argsort(pool.values, kind="stable")   # np.setdiff1d(pool, targets) is label-sorted
        if not pool.is_unique:
            by_label = by_label[np.unique(pool.values[by_label], return_index=True)[1]]

        def plan(targets):
            rows = pool.get_indexer(targets)
            if (rows < 0).any():
                missing = [t for t, r in zip(targets, rows) if r < 0]
                raise KeyError("{} not in index".format(missing[:5]))     # what .loc raises in the reference
            inside = np.zeros(len(pool), bool)
            inside[rows] = True
            cols = by_label[~inside[by_label]]         # positions of setdiff1d(pool, targets), same order
            starved = cols.size == 0
            if starved:
                cols = np.arange(len(pool))
            scores = table[np.ix_(rows, cols)]
            width = min(ntop, scores.shape[1])
            if width < scores.shape[1]:
                cand = np.argpartition(-scores, width - 1, axis=1)[:, :width]
            else:
                cand = np.broadcast_to(np.arange(scores.shape[1]), scores.shape).copy()
            rank = np.argsort(-np.take_along_axis(scores, cand, axis=1), axis=1, kind="stable")
            best = np.take_along_axis(cand, rank, axis=1)
            return pool[pd.unique(cols[best.flatten()])], starved     # first-occurrence order

        # the sub-networks are independent: plan them on the host pool, report in order
        self.predictors = []
        for net, (chosen, starved) in enumerate(_hostpar.pmap(plan, self.targets)):
            if starved:
                warnings.warn('Warning: number of target genes lower than output dim. '
                              'Consider lowering down the sub_outputdim parameter', UserWarning)
            self.predictors.append(chosen)
            print("Net {}: {} predictors, {} targets".format(net, len(chosen), len(self.targets[net])))

    def score(self, data, policy=None):
        warnings.warn("This method is deprecated. Please use model.test_metrics to measure model accuracy instead",
                      DeprecationWarning)
        estimate = self.predict(data, policy=policy)
        truth = data.loc[estimate.index, estimate.columns]
        from scipy.stats import pearsonr
        return pearsonr(estimate.values.reshape(-1), truth.values.reshape(-1))
