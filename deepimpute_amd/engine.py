"""Host-side handle for one group of sub-networks living on one GPU.

`Engine` is a thin, arithmetic-free Python face of the C ABI in include/dimn.h: it owns a
`dimn_handle` and marshals numpy arrays.  It stands where the reference holds its Keras
`Model` (deepimpute/multinet.py:226 `model = self.build(...)`, :238 `model.fit`, :253/:278
`model.predict`).  `HipEngine` binds the product library libdimn.so and raises if it is
missing -- there is no CPU fallback in the product.
"""
import ctypes as C

import os

import numpy as np

from . import _cabi
from ._cabi import Config, f32, i32, p_f32, p_f64, p_i32, p_u8


class DimnError(RuntimeError):
    pass


class Engine:
    """Generic wrapper over a bound ABI function table (see `_cabi.bind`)."""

    def __init__(self, fns, D, hidden, out_dim, batch_size=64, dropout_rate=0.2,
                 learning_rate=1e-4, beta1=0.9, beta2=0.999, eps=1e-7, loss_binary=False,
                 seed=1234, device_id=0, subnet_offset=0, activation="relu", precision="fp32"):
        self._f = fns
        self.D = [int(d) for d in D]
        self.K = len(self.D)
        self.H = int(hidden)
        self.O = int(out_dim)
        self.B = int(batch_size)
        self.subnet_offset = int(subnet_offset)
        self.cfg = Config(
            n_subnets=self.K, subnet_offset=int(subnet_offset), hidden=self.H, out_dim=self.O,
            batch_size=self.B, device_id=int(device_id), dropout_rate=float(dropout_rate),
            learning_rate=float(learning_rate), beta1=float(beta1), beta2=float(beta2),
            eps=float(eps), loss_binary=int(bool(loss_binary)), seed=int(seed), precision=_cabi.PRECISIONS[str(precision).lower()])
        self.precision = "bf16" if self.cfg.precision else "fp32"
        self._h = C.c_void_p()
        self.n_cells = 0
        self.n_train = 0
        self.n_val = 0
        Darr = i32(self.D)
        self._check(self._f["create"](C.byref(self.cfg), p_i32(Darr), C.byref(self._h)))
        self.activation = "linear" if activation is None else str(activation).lower()
        if self.activation not in _cabi.ACTIVATIONS:
            raise NotImplementedError("hidden activation %r: implemented are %s" % (activation, sorted(_cabi.ACTIVATIONS)))
        if self.activation != "relu":
            self._check(self._f["set_activation"](self._h, _cabi.ACTIVATIONS[self.activation]))

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self._f["last_error"]()
            raise DimnError("libdimn error %d: %s" % (rc, (msg or b"").decode("utf-8", "replace")))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._f["destroy"](self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data -------------------------------------------------------------
    def set_matrix(self, norm, streamed=False, with_targets=True):
        """Hand over the log1p matrix.  streamed=True (HIP engines): the matrix is streamed through pinned buffers in row
        blocks and gathered on the fly -- it never resides on the device, and gather() must not be called afterwards
        (every set_indices() first)."""
        norm = f32(norm)
        if norm.ndim != 2:
            raise ValueError("norm must be 2-D [cells, genes]")
        self.n_cells, self.n_genes = norm.shape
        self._streamed = bool(streamed)
        if streamed:
            self._check(self._f["set_matrix_streamed"](self._h, p_f32(norm), norm.shape[0], norm.shape[1], int(bool(with_targets))))
        else:
            self._check(self._f["set_matrix"](self._h, p_f32(norm), norm.shape[0], norm.shape[1]))

    def set_stream_order(self, part, parts):
        """Rank `part` of `parts` ranks reading one shared host copy of the matrix: its streamed hand-over starts that far into the row blocks."""
        if "set_stream_order" in self._f:
            self._check(self._f["set_stream_order"](self._h, int(part), int(parts)))

    def set_indices(self, k, pred_idx, targ_idx):
        pred_idx, targ_idx = i32(pred_idx), i32(targ_idx)
        if targ_idx.size != self.O:
            raise ValueError("targ_idx must have out_dim entries")
        self._check(self._f["set_indices"](self._h, k, p_i32(pred_idx), pred_idx.size, p_i32(targ_idx)))

    def gather(self, with_targets=True):
        if getattr(self, "_streamed", False):
            return                                   # set_matrix(streamed=True) has gathered already
        self._check(self._f["gather"](self._h, int(bool(with_targets))))

    def set_split(self, train_rows, val_rows):
        tr, va = i32(train_rows), i32(val_rows)
        self.n_train, self.n_val = tr.size, va.size
        self.train_rows, self.val_rows = tr, va          # (kept for callers that replay an epoch: the permutation indexes train_rows)
        self._check(self._f["set_split"](self._h, p_i32(tr), tr.size, p_i32(va), va.size))

    # -- weights ----------------------------------------------------------
    def init_weights(self, seed=None):
        self._check(self._f["init_weights"](self._h, int(self.cfg.seed if seed is None else seed)))

    def set_weights(self, k, W1, b1, W2, b2):
        W1, b1, W2, b2 = f32(W1), f32(b1), f32(W2), f32(b2)
        assert W1.shape == (self.D[k], self.H) and W2.shape == (self.H, self.O)
        assert b1.shape == (self.H,) and b2.shape == (self.O,)
        self._check(self._f["set_weights"](self._h, k, p_f32(W1), p_f32(b1), p_f32(W2), p_f32(b2)))

    def _alloc_like_weights(self, k):
        return (np.empty((self.D[k], self.H), np.float32), np.empty(self.H, np.float32),
                np.empty((self.H, self.O), np.float32), np.empty(self.O, np.float32))

    def get_weights(self, k):
        W1, b1, W2, b2 = self._alloc_like_weights(k)
        self._check(self._f["get_weights"](self._h, k, p_f32(W1), p_f32(b1), p_f32(W2), p_f32(b2)))
        return W1, b1, W2, b2

    def get_adam_state(self, k, which):
        W1, b1, W2, b2 = self._alloc_like_weights(k)
        self._check(self._f["get_adam_state"](self._h, k, int(which), p_f32(W1), p_f32(b1),
                                              p_f32(W2), p_f32(b2)))
        return W1, b1, W2, b2

    def reset_optimizer(self):
        self._check(self._f["reset_optimizer"](self._h))

    def step_count(self):
        t = C.c_int64()
        self._check(self._f["get_step_count"](self._h, C.byref(t)))
        return t.value

    # -- training / inference --------------------------------------------
    def train_step(self, rows, keep_mask=None, epoch_key=0, step_key=0, want_loss=True):
        rows = i32(rows)
        if keep_mask is not None:
            keep_mask = np.ascontiguousarray(keep_mask, dtype=np.uint8)
            assert keep_mask.shape == (self.K, rows.size, self.H)
        loss = np.empty(self.K, np.float32) if want_loss else None
        self._check(self._f["train_step"](self._h, p_i32(rows), rows.size, p_u8(keep_mask),
                                          int(epoch_key), int(step_key), p_f32(loss)))
        return loss

    def train_epoch(self, epoch, perm=None):
        perm = None if perm is None else i32(perm)
        loss = np.empty(self.K, np.float64)
        self._check(self._f["train_epoch"](self._h, int(epoch), p_i32(perm), p_f64(loss)))
        return loss

    @property
    def training_precision(self):
        """"bf16" when the second layer's training GEMMs run on the bf16 matrix cores (precision bf16 on the fused kernel), else "fp32"."""
        fn = self._f.get("training_precision")
        return "bf16" if fn is not None and fn(self._h) == 1 else "fp32"

    def val_loss(self):
        v = np.empty(self.K, np.float64)
        self._check(self._f["val_loss"](self._h, p_f64(v)))
        return v

    def fit(self, max_epochs, patience):
        lh = np.zeros(max_epochs, np.float64)
        vh = np.zeros(max_epochs, np.float64)
        n = C.c_int32()
        self._check(self._f["fit"](self._h, int(max_epochs), int(patience), p_f64(lh), p_f64(vh),
                                   C.byref(n)))
        return n.value, lh[:n.value], vh[:n.value]

    def predict(self, rows=None, n_rows=None):
        if rows is not None:
            rows = i32(rows)
            n_rows = rows.size
        elif n_rows is None:
            n_rows = self.n_cells
        out = np.empty((n_rows, self.K * self.O), np.float32)
        self._check(self._f["predict"](self._h, p_i32(rows), n_rows, p_f32(out)))
        return out

    def epoch_permutation(self, epoch, n=None, seed=None):
        n = self.n_train if n is None else n
        perm = np.empty(n, np.int32)
        self._check(self._f["epoch_permutation"](int(self.cfg.seed if seed is None else seed),
                                                 int(epoch), n, p_i32(perm)))
        return perm


class GeneralEngine(Engine):
    """Engine for ANY architecture build() accepts (reference deepimpute/multinet.py:135-162): `layers` = the hidden Dense
    layers as (neurons, activation name, dropout rate behind it); batch size and widths unrestricted; loss in
    _cabi.LOSSES.  Same face as Engine; models with several hidden layers move weights per layer."""

    def __init__(self, fns, D, layers, out_dim, batch_size=64, learning_rate=1e-4, beta1=0.9, beta2=0.999, eps=1e-7,
                 loss="wmse", seed=1234, device_id=0, subnet_offset=0, precision="fp32"):
        self._f = fns
        self.D = [int(d) for d in D]
        self.K = len(self.D)
        self.layers = [(int(n), "linear" if a is None else str(a).lower(), float(p)) for n, a, p in layers]
        # a leading (0, _, rate) entry: a Dropout layer BEFORE the first Dense layer (dropout on the inputs; include/dimn.h dimn_create_general)
        self.input_dropout = 0.0
        if self.layers and self.layers[0][0] == 0:
            self.input_dropout = self.layers[0][2]
            self.layers = self.layers[1:]
        if not self.layers:
            raise ValueError("architecture needs at least one dense layer")
        self.L = len(self.layers)
        self.H = self.layers[0][0]
        self.O = int(out_dim)
        self.B = int(batch_size)
        self.subnet_offset = int(subnet_offset)
        self.loss = str(loss).lower()
        if self.loss not in _cabi.LOSSES:
            raise NotImplementedError("loss %r: implemented are %s" % (loss, sorted(_cabi.LOSSES)))
        for _, a, _p in self.layers:
            if a not in _cabi.ACTIVATIONS:
                raise NotImplementedError("hidden activation %r: implemented are %s" % (a, sorted(_cabi.ACTIVATIONS)))
        self.cfg = Config(n_subnets=self.K, subnet_offset=int(subnet_offset), hidden=self.H, out_dim=self.O, batch_size=self.B,
                          device_id=int(device_id), dropout_rate=0.0, learning_rate=float(learning_rate), beta1=float(beta1),
                          beta2=float(beta2), eps=float(eps), loss_binary=int(self.loss == "wmse_binary"), seed=int(seed),
                          precision=_cabi.PRECISIONS[str(precision).lower()])
        self.precision = "bf16" if self.cfg.precision else "fp32"
        wire = ([(0, "linear", self.input_dropout)] if self.input_dropout > 0 else []) + self.layers
        arr = (_cabi.Layer * len(wire))(*[_cabi.Layer(n, _cabi.ACTIVATIONS[a], p) for n, a, p in wire])
        self._h = C.c_void_p()
        self.n_cells = self.n_train = self.n_val = 0
        self._check(self._f["create_general"](C.byref(self.cfg), p_i32(i32(self.D)), arr, len(wire), _cabi.LOSSES[self.loss], C.byref(self._h)))
        self.activation = self.layers[0][1]

    def layer_shape(self, k, layer):
        widths = [n for n, _, _ in self.layers] + [self.O]
        return (self.D[k] if layer == 0 else widths[layer - 1]), widths[layer]

    def set_layer_weights(self, k, layer, W, b):
        W, b = f32(W), f32(b)
        assert W.shape == self.layer_shape(k, layer) and b.shape == (W.shape[1],)
        self._check(self._f["set_layer_weights"](self._h, k, layer, p_f32(W), p_f32(b)))

    def get_layer_weights(self, k, layer, which=0):
        n_in, n_out = self.layer_shape(k, layer)
        W, b = np.empty((n_in, n_out), np.float32), np.empty(n_out, np.float32)
        self._check(self._f["get_layer_weights"](self._h, k, layer, int(which), p_f32(W), p_f32(b)))
        return W, b

    # the four-array face of the one-hidden-layer model, for code that treats every engine alike (save/load)
    def get_weights(self, k):
        out = []
        for layer in range(self.L + 1):
            out.extend(self.get_layer_weights(k, layer))
        return tuple(out)

    def set_weights(self, k, *arrays):
        assert len(arrays) == 2 * (self.L + 1)
        for layer in range(self.L + 1):
            self.set_layer_weights(k, layer, arrays[2 * layer], arrays[2 * layer + 1])

    def get_adam_state(self, k, which):
        out = []
        for layer in range(self.L + 1):
            out.extend(self.get_layer_weights(k, layer, 1 + int(which)))
        return tuple(out)

    def train_step(self, rows, keep_mask=None, epoch_key=0, step_key=0, want_loss=True):
        if keep_mask is not None:
            raise NotImplementedError("injected keep masks exist only for the one-hidden-layer engine")
        rows = i32(rows)
        loss = np.empty(self.K, np.float32) if want_loss else None
        if "train_step_general" in self._f:
            self._check(self._f["train_step_general"](self._h, p_i32(rows), rows.size, int(epoch_key), int(step_key), p_f32(loss)))
        else:
            self._check(self._f["train_step"](self._h, p_i32(rows), rows.size, None, int(epoch_key), int(step_key), p_f32(loss)))
        return loss


class FrameMismatch(Exception):
    """impute_finish(raw=None, observed=frame): `frame` turned out not to be the count matrix the device holds."""


class HipEngine(Engine):
    """Engine on libdimn.so (hand-written HIP kernels for gfx950).  Raises ImportError-like
    `DimnError` when the library is not built: the product never falls back to a CPU path."""

    def __init__(self, D, hidden, out_dim, **kw):
        from ._lib import load
        super().__init__(load(), D, hidden, out_dim, **kw)

    def predict_device(self, rows=None, n_rows=None):
        if rows is not None:
            rows = i32(rows)
            n_rows = rows.size
        elif n_rows is None:
            n_rows = self.n_cells
        ptr = C.c_void_p()
        self._check(self._f["predict_device"](self._h, p_i32(rows), n_rows, C.byref(ptr)))
        return ptr.value

    def val_metrics(self):
        """(count, Sx, Sy, Sxx, Syy, Sxy, S(x-y)^2) over the positive validation targets (include/dimn.h)."""
        out = np.zeros(7, np.float64)
        self._check(self._f["val_metrics"](self._h, p_f64(out)))
        return out

    restore_epilogue = True      # (tools/finish_ab.py sets it False to time the dense epilogue on the same box)

    def impute_finish(self, raw, gene_off, gene_slot, policy, ceiling, from_gathered=False, observed=None):
        """predict()'s post-processing on the device over the last predict_device() result (include/dimn.h);
        raw [cells, genes] float64 -> the finished [cells, genes] float64 matrix.  raw = None: the observed counts are the
        resident matrix of set_matrix_counts(); with policy "restore" and `observed` (the caller's C-ordered float64 frame of
        those counts) only the zero entries come back over PCIe (dimn_impute_finish_restore) -- a frame that turns out not to be
        the resident matrix falls through to the ordinary call."""
        if raw is None:
            shape = (self.n_cells, self.n_genes)
        else:
            raw = np.ascontiguousarray(raw, dtype=np.float64)
            shape = raw.shape
        gene_off, gene_slot = i32(gene_off), i32(gene_slot)
        out = np.empty(shape, np.float64)
        self.last_observed_checksum = None           # set by the restore form: the checksum of `observed` as it was read
        obs_dtype = _cabi.count_dtype(observed)          # float64 or int64, C-ordered
        if raw is None and policy == "restore" and self.restore_epilogue and obs_dtype is not None and observed.shape == shape:
            cs = C.c_uint64(0)
            rc = self._f["impute_finish_restore"](self._h, observed.ctypes.data, obs_dtype, shape[0], shape[1], p_i32(gene_off), p_i32(gene_slot),
                                                  float(ceiling), int(bool(from_gathered)), p_f64(out), C.byref(cs))
            if rc == 0:
                self.last_observed_checksum = int(cs.value)
                return out
            if rc == -3:
                # DIMN_ERR_STATE: `observed` is not the matrix the device holds.  The dense epilogue over the RESIDENT counts would finish a
                # frame the caller did not pass (8 GB of device-to-host copy at 50k x 20k, thrown away): say so instead -- predict() uploads
                # the frame it was given and runs the ordinary sequence once.
                raise FrameMismatch("the frame is not the count matrix resident on the device")
            self._check(rc)
        code = {None: 0, "restore": 1, "max": 2}.get(policy, 0)
        self._check(self._f["impute_finish"](self._h, p_f64(raw), shape[0], shape[1], p_i32(gene_off), p_i32(gene_slot),
                                             code, float(ceiling), int(bool(from_gathered)), p_f64(out)))
        return out

    def set_matrix_counts(self, counts):
        """The log1p matrix of this engine from a resident count matrix (_counts.DeviceCounts): log1p through numpy's own table,
        on the device; gather() follows as after set_matrix().  The engine remembers the counts (impute_finish(raw=None))."""
        lut = counts.log1p_table()
        self._check(self._f["set_matrix_counts"](self._h, counts.handle, p_f32(lut), lut.size))
        self.n_cells, self.n_genes = counts.n, counts.g
        self._streamed = False
        self._dev_counts = counts                # keeps the device matrix alive as long as this engine may read it

    def path_info(self):
        """The kernels dimn_create chose for this handle (include/dimn.h dimn_path_info), as a dict."""
        out = np.zeros(8, np.int32)
        self._check(self._f["path_info"](self._h, p_i32(out)))
        keys = ("path", "resident_groups", "resident_splits", "mid_fused", "mid_slices", "mid_kernel", "train_bf16", "first_layer")
        d = dict(zip(keys, (int(x) for x in out)))
        d["mid_keep"] = d["mid_kernel"]              # (the key's name until round 4, when 2 = the tile pipeline joined 1 / 0 = k_mid_fused with / without W2 kept in LDS)
        d["path"] = ("streaming", "resident", "general")[d["path"]]
        return d

    def synchronize(self):
        self._check(self._f["synchronize"](self._h))

    def set_profiling(self, on):
        self._check(self._f["set_profiling"](self._h, int(bool(on))))

    def get_timers(self, reset=True):
        out = np.zeros(8, np.float64)
        self._check(self._f["get_timers"](self._h, p_f64(out), int(bool(reset))))
        return out

    # RCCL -------------------------------------------------------------
    def comm_unique_id(self):
        buf = np.zeros(_cabi.COMM_ID_BYTES, np.uint8)
        self._check(self._f["comm_unique_id"](p_u8(buf)))
        return buf

    def comm_init(self, uid, n_ranks, rank):
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        self._check(self._f["comm_init"](self._h, p_u8(uid), int(n_ranks), int(rank)))

    def comm_info(self):
        """(ranks, rank) as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        out = np.zeros(2, np.int32)
        self._check(self._f["comm_info"](self._h, p_i32(out)))
        return int(out[0]), int(out[1])

    def comm_allreduce_sum(self, vec):
        v = np.ascontiguousarray(vec, dtype=np.float64).copy()
        self._check(self._f["comm_allreduce_sum"](self._h, p_f64(v), v.size))
        return v

    def comm_gather_predictions(self, n_rows, counts, root=0, is_root=False):
        counts = i32(counts)
        out = np.empty((n_rows, int(counts.sum()) * self.O), np.float32) if is_root else None
        self._check(self._f["comm_gather_predictions"](self._h, n_rows, p_i32(counts), int(root),
                                                       p_f32(out)))
        return out

    def comm_destroy(self):
        self._check(self._f["comm_destroy"](self._h))

    @staticmethod
    def gather_loopback(engines, n_rows, root=0, want_host=True):
        """dimn_comm_gather_loopback: the engines (all on one GPU, each after predict_device() over n_rows) play the ranks of a
        sharded job; returns root's gathered [n_rows, sum(K_r) * O] matrix (it also stays in root's HBM for impute_finish)."""
        handles = (C.c_void_p * len(engines))(*[e._h for e in engines])
        counts = i32([e.K for e in engines])
        out = np.empty((n_rows, int(counts.sum()) * engines[root].O), np.float32) if want_host else None
        engines[root]._check(engines[root]._f["comm_gather_loopback"](handles, len(engines), int(n_rows), p_i32(counts), int(root), p_f32(out)))
        return out


class HipGeneralEngine(GeneralEngine, HipEngine):
    """GeneralEngine on libdimn.so (dimn_create_general: batched fp32-MFMA GEMMs per layer, dimn_general.h)."""

    def __init__(self, D, layers, out_dim, **kw):
        from ._lib import load
        GeneralEngine.__init__(self, load(), D, layers, out_dim, **kw)
