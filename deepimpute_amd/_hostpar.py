"""Thread-pool helpers for the host-side planning around the GPU path.

The reference does its planning (per-gene statistics, log1p, predictor selection, post-processing of the
predictions; deepimpute/multinet.py:20-34, 191-216, 282-310) with whole-matrix pandas/numpy calls on one
core.  Every one of those is independent per gene column or per cell row, so here the SAME pandas/numpy
routine runs on column / row blocks from a thread pool (numpy releases the GIL inside its loops): the
per-element arithmetic -- and therefore every value, to the bit -- is what the whole-matrix call gives
(tests/test_shell.py checks that), only the wall time changes.  Nothing here touches the GPU.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pandas as pd


def n_workers():
    n = int(os.environ.get("DIMN_HOST_THREADS", "0") or 0)
    if n <= 0:
        n = min(32, os.cpu_count() or 1)
    return max(1, n)


def spans(total, block):
    """[(start, stop)] covering range(total) in pieces of `block`."""
    block = max(1, int(block))
    return [(a, min(a + block, total)) for a in range(0, total, block)]


def pmap(fn, items):
    """[fn(x) for x in items], evaluated on the pool, results in order."""
    items = list(items)
    if len(items) <= 1 or n_workers() == 1:
        return [fn(x) for x in items]
    with ThreadPoolExecutor(n_workers()) as pool:
        return list(pool.map(fn, items))


def _block_for(frame_or_shape, axis, target_bytes=32 << 20):
    shape = getattr(frame_or_shape, "shape", frame_or_shape)
    other = shape[1 - axis]
    return max(16, int(target_bytes // max(1, 8 * other)))


_pandas_order_ok = None
_pandas_order_lock = __import__("threading").RLock()      # (re-entrant: the probe itself calls _native_col_stats, which may ask again)


def pandas_order_holds():
    """One-time self-check of what the native gene statistics (host: dimn_col_stats*, device: dimn_counts_gene_stats) assume about
    THIS pandas / numpy: DataFrame.mean() a running sum down the column, DataFrame.var() numpy's pairwise reduction over 8192-element
    buffer chunks (no bottleneck, csrc/dimn_hoststats.h).  A small count matrix crossing one chunk boundary must come out of the
    native routine equal to pandas TO THE BIT; otherwise the native paths are switched off for the process (pandas computes the
    statistics: the gene ranking of fit() then follows the installed library, as the reference's does) and a warning says so."""
    global _pandas_order_ok
    if _pandas_order_ok is not None and _pandas_order_ok != "probing":
        return _pandas_order_ok
    with _pandas_order_lock:                         # other threads (pmap workers, the upload / correlation helpers) wait for the verdict
        if _pandas_order_ok == "probing":
            return True                              # only the probing thread itself gets here (RLock): its own call of the native routine
        if _pandas_order_ok is None:
            _pandas_order_ok = "probing"
            verdict = True
            try:
                rng = np.random.default_rng(20240607)
                probe = pd.DataFrame(rng.poisson(rng.gamma(2.0, 3.0, size=(8200, 6))).astype(np.float64))
                got = _native_col_stats(probe.values)
                if got is not None:
                    verdict = bool(np.array_equal(got[0], probe.mean().values) and np.array_equal(got[1], probe.var().values))
                    if not verdict:
                        import warnings
                        warnings.warn("deepimpute_amd: the native gene statistics differ from this pandas' DataFrame.mean()/var() at ulp level "
                                      "(pandas %s, numpy %s); using pandas' own reductions" % (pd.__version__, np.__version__), RuntimeWarning)
            finally:
                _pandas_order_ok = verdict
        return _pandas_order_ok


def _native_col_stats(values, want_var=True):
    """(mean, var, max) of the columns of a C-ordered float64 matrix through libdimn's host routine (dimn_col_stats:
    pandas' order of operations, multi-threaded), or None when the library is not built / a NaN is present."""
    if values.dtype != np.float64 or values.ndim != 2 or not values.flags.c_contiguous or values.shape[0] < 2 or values.shape[1] < 1:
        return None
    if not pandas_order_holds():
        return None
    try:
        from . import _cabi, _lib
        fn = _lib.load()["col_stats"]
    except (ImportError, OSError, KeyError):
        return None
    import ctypes as C
    n, g = values.shape
    mean, var = np.empty(g, np.float64), np.empty(g, np.float64) if want_var else None
    vmax, has_nan = C.c_double(), C.c_int32()
    rc = fn(_cabi.p_f64(values), n, g, g, _cabi.p_f64(mean), _cabi.p_f64(var), C.byref(vmax), C.byref(has_nan),
            int(os.environ.get("DIMN_HOST_THREADS", "0") or 0))
    if rc != 0 or has_nan.value:
        return None
    return mean, var, vmax.value


def col_stats_first(values):
    """First sweep of the gene statistics (dimn_col_stats_first): dict(mean, avg, cmin, cmax, vmax) of a NaN-free C-ordered
    float64 matrix, or None (library missing, NaN present, another dtype / layout)."""
    if os.environ.get("DIMN_HOST_STATS", "1") == "0" or not isinstance(values, np.ndarray) or values.dtype != np.float64 or values.ndim != 2 \
            or not values.flags.c_contiguous or values.shape[0] < 2 or values.shape[1] < 1 or not pandas_order_holds():
        return None
    try:
        from . import _cabi, _lib
        fn = _lib.load()["col_stats_first"]
    except (ImportError, OSError, KeyError):
        return None
    import ctypes as C
    n, g = values.shape
    out = {k: np.empty(g, np.float64) for k in ("mean", "avg", "cmin", "cmax")}
    vmax, has_nan = C.c_double(), C.c_int32()
    rc = fn(_cabi.p_f64(values), n, g, g, _cabi.p_f64(out["mean"]), _cabi.p_f64(out["avg"]), _cabi.p_f64(out["cmin"]), _cabi.p_f64(out["cmax"]),
            C.byref(vmax), C.byref(has_nan), int(os.environ.get("DIMN_HOST_THREADS", "0") or 0))
    if rc != 0 or has_nan.value:
        return None
    out["vmax"] = vmax.value
    return out


def col_stats_var(values, avg):
    """Second sweep (dimn_col_stats_var): DataFrame.var() of the columns from the averages of col_stats_first, to the bit."""
    from . import _cabi, _lib
    n, g = values.shape
    var = np.empty(g, np.float64)
    if _lib.load()["col_stats_var"](_cabi.p_f64(values), n, g, g, _cabi.p_f64(avg), _cabi.p_f64(var), int(os.environ.get("DIMN_HOST_THREADS", "0") or 0)) != 0:
        raise RuntimeError("dimn_col_stats_var failed")
    return var


def column_var_mean(frame):
    """(frame.var(), frame.mean()): libdimn's host routine in pandas' order of operations (bit-identical, tests/test_shell.py)
    for a NaN-free float64 frame, else pandas' own reductions on column blocks."""
    values = frame.values if hasattr(frame, "values") else None
    got = _native_col_stats(values) if values is not None and os.environ.get("DIMN_HOST_STATS", "1") != "0" else None
    if got is not None:
        mean, var, _ = got
        return pd.Series(var, index=frame.columns), pd.Series(mean, index=frame.columns)

    def one(ab):
        part = frame.iloc[:, ab[0]:ab[1]]
        return part.var(), part.mean()
    parts = pmap(one, spans(frame.shape[1], _block_for(frame, 1)))
    if not parts:
        return frame.var(), frame.mean()
    return pd.concat([p[0] for p in parts]), pd.concat([p[1] for p in parts])


def log1p_float32(frame):
    """np.log1p(frame).astype(np.float32) as a frame (multinet.py:216), by row blocks."""
    values = frame.values
    out = np.empty(values.shape, np.float32)

    def one(ab):
        out[ab[0]:ab[1]] = np.log1p(values[ab[0]:ab[1]])      # log1p in the input dtype, then the cast
    pmap(one, spans(values.shape[0], _block_for(values, 0)))
    return pd.DataFrame(out, index=frame.index, columns=frame.columns, copy=False)


def zero_nans_inplace(square):
    """square[np.isnan(square)] = 0 by row blocks (== DataFrame.fillna(0) on a float matrix)."""
    def one(ab):
        part = square[ab[0]:ab[1]]
        part[np.isnan(part)] = 0
    pmap(one, spans(square.shape[0], _block_for(square, 0)))
    return square


def take_columns(values, positions):
    """values[:, positions] (a C-ordered copy) by row blocks."""
    positions = np.asarray(positions)
    out = np.empty((values.shape[0], len(positions)), values.dtype)

    def one(ab):
        np.take(values[ab[0]:ab[1]], positions, axis=1, out=out[ab[0]:ab[1]])
    pmap(one, spans(values.shape[0], _block_for(out, 0)))
    return out


def matrix_max(values):
    """values.max(): libdimn's threaded host pass for a NaN-free float64 matrix, else numpy by row blocks (NaN propagates)."""
    got = _native_col_stats(values, want_var=False) if os.environ.get("DIMN_HOST_STATS", "1") != "0" else None
    if got is not None:
        return np.float64(got[2])
    parts = pmap(lambda ab: values[ab[0]:ab[1]].max(), spans(values.shape[0], _block_for(values, 0)))
    return np.max(parts) if parts else values.max()


def as_float64(values):
    """np.ascontiguousarray(values, dtype=np.float64); the conversion (integer or float32 counts) by row blocks."""
    if values.dtype == np.float64 and values.flags.c_contiguous:
        return values
    out = np.empty(values.shape, np.float64)

    def one(ab):
        out[ab[0]:ab[1]] = values[ab[0]:ab[1]]
    pmap(one, spans(values.shape[0], _block_for(out, 0)))
    return out
