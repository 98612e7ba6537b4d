"""The raw count matrix resident on the GPU (include/dimn.h, dimn_counts_*): uploaded ONCE as float32, then read in place by
the correlation / predictor selection, by the log1p hand-over of the engine (through a numpy-computed table) and by predict()'s
restore / max step.  The reference passes the same frame through numpy four times (deepimpute/multinet.py:191, 20-34, 216,
292-303); at 50k x 20k each pass -- and each 4-8 GB host-to-device copy of it -- costs a large part of a second next to two
seconds of training.  Only for what the tool is specified for: a matrix of raw counts (non-negative integers); anything else
makes `DeviceCounts.try_create` return None and MultiNet keeps its host path.  Nothing here does arithmetic on the data except
`log1p_table`, which is numpy's own log1p."""
import ctypes as C

import numpy as np

from . import _cabi


class DeviceCounts:
    def __init__(self, handle, n, g, vmax, checksum, device_id):
        self.handle, self.n, self.g, self.vmax, self.checksum, self.device_id = handle, int(n), int(g), float(vmax), int(checksum), int(device_id)

    @staticmethod
    def try_create(values, device_id=0):
        """Upload `values` ([cells, genes] float64 or int64, C-ordered) or return None: library missing, no GPU, or values that are
        not counts (the library checks every element on the way)."""
        dtype = _cabi.count_dtype(values)
        if dtype is None or values.size == 0:
            return None
        try:
            from . import _lib
            fns = _lib.load()
        except (ImportError, OSError):
            return None
        h, vmax, cs = C.c_void_p(), C.c_double(), C.c_uint64()
        rc = fns["counts_create_typed"](int(device_id), values.ctypes.data, dtype, values.shape[0], values.shape[1], C.byref(vmax), C.byref(cs), C.byref(h))
        if rc != 0:
            return None
        return DeviceCounts(h, values.shape[0], values.shape[1], vmax.value, cs.value, device_id)

    def shape_matches(self, values):
        """The cheap half of matches(): a C-ordered float64 / int64 frame of the uploaded shape."""
        return self.handle is not None and _cabi.count_dtype(values) is not None and values.shape == (self.n, self.g)

    def matches(self, values):
        """True when `values` is, bit for bit, the matrix that was uploaded (one threaded host pass: a position-dependent
        checksum of the float64 bit patterns -- of (double)v for an int64 frame: the same numbers match whatever type carries them)."""
        if not self.shape_matches(values):
            return False
        from . import _lib
        cs = C.c_uint64()
        if _lib.load()["counts_checksum_typed"](values.ctypes.data, _cabi.count_dtype(values), self.n, self.g, C.byref(cs)) != 0:
            return False
        return cs.value == self.checksum

    def log1p_table(self):
        """float32(log1p(v)) for v = 0 .. max count, computed by numpy exactly as the reference computes
        np.log1p(raw).astype(np.float32) element by element (multinet.py:216-217)."""
        return np.log1p(np.arange(int(self.vmax) + 1, dtype=np.float64)).astype(np.float32)

    def select_predictors(self, pool_cols, targ_pos, col_rank, ntop):
        from . import _lib
        fns = _lib.load()
        pool_cols, targ_pos, col_rank = _cabi.i32(pool_cols), _cabi.i32(targ_pos), _cabi.i32(col_rank)
        K, O = targ_pos.shape
        picks = np.empty((K, O, int(ntop)), np.int32)
        rc = fns["counts_select_predictors"](self.handle, _cabi.p_i32(pool_cols), pool_cols.size, _cabi.p_i32(targ_pos), K, O, _cabi.p_i32(col_rank),
                                             int(ntop), _cabi.p_i32(picks))
        if rc != 0:
            raise RuntimeError("dimn_counts_select_predictors: " + fns["last_error"]().decode("utf-8", "replace"))
        return picks

    def corr(self, pool_cols):
        """|corr| of the pool columns, left on the device for topk() (dimn_counts_corr)."""
        from . import _lib
        fns = _lib.load()
        pool_cols = _cabi.i32(pool_cols)
        if fns["counts_corr"](self.handle, _cabi.p_i32(pool_cols), pool_cols.size) != 0:
            raise RuntimeError("dimn_counts_corr: " + fns["last_error"]().decode("utf-8", "replace"))

    def corr_drop(self):
        """Free what corr() left on the device when the selection takes another path (pool^2 * 8 bytes)."""
        if self.handle is not None and self.handle:
            from . import _lib
            _lib.load()["counts_corr_drop"](self.handle)
        self.corr_ready = False

    def corr_read(self, pool_n):
        """(tests / diagnostics) the |corr| matrix corr() left on the device."""
        from . import _lib
        fns = _lib.load()
        out = np.empty((int(pool_n), int(pool_n)), np.float64)
        if fns["counts_corr_read"](self.handle, _cabi.p_f64(out), int(pool_n)) != 0:
            raise RuntimeError("dimn_counts_corr_read: " + fns["last_error"]().decode("utf-8", "replace"))
        return out

    def gene_stats(self):
        """dict(mean, var, cmin, cmax) of the columns: DataFrame.mean() / .var() to the bit (dimn_counts_gene_stats), computed
        from the resident copy; `vmax` is the matrix maximum the upload found."""
        from . import _lib
        fns = _lib.load()
        out = {k: np.empty(self.g, np.float64) for k in ("mean", "var", "cmin", "cmax")}
        if fns["counts_gene_stats"](self.handle, _cabi.p_f64(out["mean"]), _cabi.p_f64(out["var"]), _cabi.p_f64(out["cmin"]), _cabi.p_f64(out["cmax"])) != 0:
            raise RuntimeError("dimn_counts_gene_stats: " + fns["last_error"]().decode("utf-8", "replace"))
        out["vmax"] = self.vmax
        return out

    def topk(self, targ_pos, col_rank, ntop):
        from . import _lib
        fns = _lib.load()
        targ_pos, col_rank = _cabi.i32(targ_pos), _cabi.i32(col_rank)
        K, O = targ_pos.shape
        picks = np.empty((K, O, int(ntop)), np.int32)
        if fns["counts_topk"](self.handle, _cabi.p_i32(targ_pos), K, O, _cabi.p_i32(col_rank), int(ntop), _cabi.p_i32(picks)) != 0:
            raise RuntimeError("dimn_counts_topk: " + fns["last_error"]().decode("utf-8", "replace"))
        return picks

    def close(self):
        if self.handle is not None and self.handle:
            from . import _lib
            _lib.load()["counts_destroy"](self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
