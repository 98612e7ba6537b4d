"""ctypes face of the CPU oracle (oracle/dimo.c).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by deepimpute_amd."""
import ctypes as C
import os
import subprocess

from deepimpute_amd import _cabi
from deepimpute_amd.engine import Engine, GeneralEngine

_HERE = os.path.dirname(os.path.abspath(__file__))
_cache = {}


def build(quiet=True):
    subprocess.check_call(["make", "-C", _HERE] + (["-s"] if quiet else []))


def _load(fp64=False, path=None):
    name = path or os.path.join(_HERE, "libdimo64.so" if fp64 else "libdimo.so")
    if name not in _cache:
        if not os.path.exists(name):
            build()
        lib = C.CDLL(name)
        fns = _cabi.bind(lib, "dimo_", gpu=False)
        lib.dimo_real_bytes.restype = C.c_int
        assert lib.dimo_real_bytes() == (8 if fp64 else 4)
        _cache[name] = fns
    return _cache[name]


class OracleEngine(Engine):
    """Same Python face as HipEngine, running the plain-loop C restatement on the CPU."""

    def __init__(self, D, hidden, out_dim, fp64=False, lib_path=None, infer_bf16=False, train_bf16=False, **kw):
        super().__init__(_load(fp64, lib_path), D, hidden, out_dim, **kw)
        name = lib_path or os.path.join(_HERE, "libdimo64.so" if fp64 else "libdimo.so")
        self._lib_name = name
        if infer_bf16:               # restates k_predict_bf16: inference / validation GEMM operands rounded to bfloat16
            lib = C.CDLL(name)
            lib.dimo_set_inference_bf16.argtypes = [C.c_void_p, C.c_int32]
            lib.dimo_set_inference_bf16(self._h, 1)
        if train_bf16:               # 1 / True restates k_mid_fused<KEEP, BF>: the second layer's three training GEMMs on bf16 operands;
            lib = C.CDLL(name)       # 2 restates k_epoch_resident<.., BF>: the first layer's two training GEMMs as well
            lib.dimo_set_training_bf16.argtypes = [C.c_void_p, C.c_int32]
            lib.dimo_set_training_bf16(self._h, int(train_bf16))


    def last_dz(self, k, b_act):
        """Test instrument (dimo_debug_last_dz): (dL/dz, z) of sub-net k over the last train_step's batch, float64 [b_act, O]."""
        import numpy as np
        lib = C.CDLL(self._lib_name)
        lib.dimo_debug_last_dz.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        dz, z = np.empty((b_act, self.O), np.float64), np.empty((b_act, self.O), np.float64)
        if lib.dimo_debug_last_dz(self._h, int(k), int(b_act), dz.ctypes.data_as(C.POINTER(C.c_double)), z.ctypes.data_as(C.POINTER(C.c_double))) != 0:
            raise ValueError("last_dz: bad argument")
        return dz, z

    def invert_gate(self, k, epoch, step, b, unit):
        """Test instrument (oracle/dimo.c dimo_invert_gate): take the relu gate of (sub-net k, epoch, step, batch position b,
        hidden unit) on the other side of zero -- for pre-activations the fp64 replay shows to be at fp32 noise level."""
        lib = C.CDLL(self._lib_name)
        lib.dimo_invert_gate.argtypes = [C.c_void_p] + [C.c_int32] * 5
        if lib.dimo_invert_gate(self._h, int(k), int(epoch), int(step), int(b), int(unit)) != 0:
            raise ValueError("invert_gate: bad argument or more than 8 inversions")


def _bind_general_oracle(lib):
    """Function table of the general CPU oracle (prefix dimog_; `create` there is the general constructor)."""
    names = ["destroy", "set_matrix", "set_indices", "set_split", "init_weights", "get_step_count", "train_step_general",
             "train_epoch", "val_loss", "fit", "predict", "epoch_permutation"]
    fns = {}
    for name in names:
        real = "train_step" if name == "train_step_general" else name
        fn = getattr(lib, "dimog_" + real)
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float)] if name == "train_step_general" else _cabi.SIGNATURES[name]
        fn.restype = C.c_int
        fns[name] = fn
    for name, args in (("create", _cabi.GENERAL["create_general"]), ("set_layer_weights", _cabi.GENERAL["set_layer_weights"]),
                       ("get_layer_weights", _cabi.GENERAL["get_layer_weights"])):
        fn = getattr(lib, "dimog_" + name)
        fn.argtypes = args
        fn.restype = C.c_int
        fns["create_general" if name == "create" else name] = fn
    fns["gather"] = lambda h, w: 0
    le = lib.dimog_last_error
    le.argtypes = []
    le.restype = C.c_char_p
    fns["last_error"] = le
    return fns


def _load_general(fp64=False):
    name = os.path.join(_HERE, "libdimo_gen64.so" if fp64 else "libdimo_gen.so")
    if name not in _cache:
        if not os.path.exists(name):
            build()
        lib = C.CDLL(name)
        lib.dimog_real_bytes.restype = C.c_int
        assert lib.dimog_real_bytes() == (8 if fp64 else 4)
        _cache[name] = _bind_general_oracle(lib)
    return _cache[name]


class GeneralOracleEngine(GeneralEngine):
    """The general CPU oracle (oracle/dimo_general.c) behind the GeneralEngine face."""

    def __init__(self, D, layers, out_dim, fp64=False, **kw):
        kw.pop("device_id", None)
        super().__init__(_load_general(fp64), D, layers, out_dim, **kw)
