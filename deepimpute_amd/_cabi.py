"""ctypes view of the C ABI declared in include/dimn.h.

`bind(lib, prefix)` attaches argtypes/restype for every entry point of the ABI to a loaded
shared library.  The product binds `libdimn.so` with prefix ``dimn_`` (see `_lib.py`);
the test-suite binds the CPU oracle, which exports the same signatures under ``dimo_``.
Nothing in this module performs arithmetic.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 9
MAX_BATCH = 64
COMM_ID_BYTES = 128
# hidden-layer activations the kernels implement (ids = DIMN_ACT_* of include/dimn.h; Keras names)
ACTIVATIONS = {"relu": 0, "linear": 1, "sigmoid": 2, "tanh": 3, "elu": 4, "softplus": 5,
               "selu": 6, "softsign": 7, "swish": 8, "gelu": 9, "exponential": 10, "hard_sigmoid": 11}


class Config(C.Structure):
    """struct dimn_config (include/dimn.h); mirrors MultiNet.build()'s arguments
    (reference deepimpute/multinet.py:126-167)."""
    _fields_ = [
        ("n_subnets", C.c_int32),
        ("subnet_offset", C.c_int32),
        ("hidden", C.c_int32),
        ("out_dim", C.c_int32),
        ("batch_size", C.c_int32),
        ("device_id", C.c_int32),
        ("dropout_rate", C.c_float),
        ("learning_rate", C.c_float),
        ("beta1", C.c_float),
        ("beta2", C.c_float),
        ("eps", C.c_float),
        ("loss_binary", C.c_int32),
        ("seed", C.c_uint64),
        ("precision", C.c_int32),
    ]


class Layer(C.Structure):
    """struct dimn_layer (include/dimn.h): one hidden Dense layer + the rate of the Dropout layer behind it."""
    _fields_ = [("neurons", C.c_int32), ("activation", C.c_int32), ("dropout_rate", C.c_float)]


PRECISIONS = {"fp32": 0, "f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}
LOSSES = {"wmse": 0, "wmse_binary": 1, "mean_squared_error": 2, "mse": 2, "mean_absolute_error": 3, "mae": 3,
          # round 5: the other element-wise keras.losses names (and their Keras aliases)
          "mean_squared_logarithmic_error": 4, "msle": 4, "logcosh": 5, "log_cosh": 5, "huber": 6, "huber_loss": 6, "poisson": 7}

_H = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_pf = C.POINTER(C.c_float)
_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pu8 = C.POINTER(C.c_uint8)

# name -> argtypes (restype int unless listed in _RESTYPE)
SIGNATURES = {
    "create": [C.POINTER(Config), _pi, C.POINTER(_H)],
    "destroy": [_H],
    "set_matrix": [_H, _pf, _i64, _i64],
    "set_indices": [_H, _i32, _pi, _i32, _pi],
    "gather": [_H, _i32],
    "set_split": [_H, _pi, _i64, _pi, _i64],
    "init_weights": [_H, C.c_uint64],
    "set_weights": [_H, _i32, _pf, _pf, _pf, _pf],
    "get_weights": [_H, _i32, _pf, _pf, _pf, _pf],
    "get_adam_state": [_H, _i32, _i32, _pf, _pf, _pf, _pf],
    "set_activation": [_H, _i32],
    "reset_optimizer": [_H],
    "get_step_count": [_H, C.POINTER(_i64)],
    "train_step": [_H, _pi, _i32, _pu8, _i32, _i32, _pf],
    "train_epoch": [_H, _i32, _pi, _pd],
    "val_loss": [_H, _pd],
    "fit": [_H, _i32, _i32, _pd, _pd, _pi],
    "predict": [_H, _pi, _i64, _pf],
    "epoch_permutation": [C.c_uint64, _i32, _i64, _pi],
}
# the general path (any architecture / batch size / loss): include/dimn.h dimn_create_general & co; the general
# oracle (oracle/dimo_general.c) exports the same under dimog_
GENERAL = {
    "create_general": [C.POINTER(Config), _pi, C.POINTER(Layer), _i32, _i32, C.POINTER(_H)],
    "set_layer_weights": [_H, _i32, _i32, _pf, _pf],
    "get_layer_weights": [_H, _i32, _i32, _i32, _pf, _pf],
}
# entry points only the GPU library has
GPU_ONLY = {
    "abi_version": [],
    "device_count": [C.POINTER(C.c_int32)],
    "release_cached_memory": [],
    "cached_memory_info": [C.POINTER(C.c_int64)],
    "warm_up": [_i32],
    "set_matrix_streamed": [_H, _pf, _i64, _i64, _i32],
    "set_stream_order": [_H, _i32, _i32],
    "predict_device": [_H, _pi, _i64, C.POINTER(C.c_void_p)],
    "synchronize": [_H],
    "get_timers": [_H, _pd, _i32],
    "training_precision": [_H],
    "set_profiling": [_H, _i32],
    "path_info": [_H, _pi],
    "comm_unique_id": [_pu8],
    "comm_init": [_H, _pu8, _i32, _i32],
    "comm_info": [_H, _pi],
    "comm_allreduce_sum": [_H, _pd, _i32],
    "comm_gather_predictions": [_H, _i64, _pi, _i32, _pf],
    "comm_destroy": [_H],
    "comm_gather_loopback": [C.POINTER(_H), _i32, _i64, _pi, _i32, _pf],
    "abs_corrcoef": [_i32, _pd, _i64, _i64, _pd],
    "val_metrics": [_H, _pd],
    "impute_finish": [_H, _pd, _i64, _i64, _pi, _pi, _i32, C.c_double, _i32, _pd],
    "impute_finish_restore": [_H, C.c_void_p, _i32, _i64, _i64, _pi, _pi, C.c_double, _i32, _pd, C.POINTER(C.c_uint64)],
    "csv_scan": [C.c_char_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "csv_read": [C.c_char_p, _i64, _i64, C.POINTER(_i64), C.c_char_p, _i64],
    "csv_write": [C.c_char_p, _pd, _i64, _i64, C.c_char_p, C.c_char_p, C.c_char_p],
    "counts_create": [_i32, _pd, _i64, _i64, _pd, C.POINTER(C.c_uint64), C.POINTER(_H)],
    "counts_create_typed": [_i32, C.c_void_p, _i32, _i64, _i64, _pd, C.POINTER(C.c_uint64), C.POINTER(_H)],
    "counts_checksum": [_pd, _i64, _i64, C.POINTER(C.c_uint64)],
    "counts_checksum_typed": [C.c_void_p, _i32, _i64, _i64, C.POINTER(C.c_uint64)],
    "counts_destroy": [_H],
    "counts_select_predictors": [_H, _pi, _i64, _pi, _i32, _i32, _pi, _i32, _pi],
    "counts_corr": [_H, _pi, _i64],
    "counts_topk": [_H, _pi, _i32, _i32, _pi, _i32, _pi],
    "counts_corr_read": [_H, _pd, _i64],
    "counts_corr_drop": [_H],
    "counts_gene_stats": [_H, _pd, _pd, _pd, _pd],
    "set_matrix_counts": [_H, _H, _pf, _i64],
    "col_stats_first": [_pd, _i64, _i64, _i64, _pd, _pd, _pd, _pd, _pd, _pi, _i32],
    "col_stats_var": [_pd, _i64, _i64, _i64, _pd, _pd, _i32],
    "col_stats": [_pd, _i64, _i64, _i64, _pd, _pd, _pd, _pi, _i32],
    "select_predictors": [_i32, _pd, _i64, _i64, _pi, _i32, _i32, _pi, _i32, _pi],
}


def bind(lib, prefix, gpu=False):
    """Attach signatures; raises AttributeError naming the first missing symbol."""
    table = dict(SIGNATURES)
    if gpu:
        table.update(GPU_ONLY)
        table.update(GENERAL)
    fns = {}
    for name, argtypes in table.items():
        fn = getattr(lib, prefix + name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
        fns[name] = fn
    le = getattr(lib, prefix + "last_error")
    le.argtypes = []
    le.restype = C.c_char_p
    fns["last_error"] = le
    return fns


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def p_f32(a):
    return None if a is None else a.ctypes.data_as(_pf)


def p_f64(a):
    return None if a is None else a.ctypes.data_as(_pd)


def p_i32(a):
    return None if a is None else a.ctypes.data_as(_pi)


def p_u8(a):
    return None if a is None else a.ctypes.data_as(_pu8)


# element types a host count matrix may have (include/dimn.h DIMN_DTYPE_*): float64, or int64 (what pd.read_csv makes of a count CSV)
COUNT_DTYPES = {np.dtype(np.float64): 0, np.dtype(np.int64): 1}


def count_dtype(a):
    """DIMN_DTYPE_* of a C-ordered 2-D float64 / int64 array, else None."""
    if isinstance(a, np.ndarray) and a.ndim == 2 and a.flags.c_contiguous and a.dtype in COUNT_DTYPES:
        return COUNT_DTYPES[a.dtype]
    return None
