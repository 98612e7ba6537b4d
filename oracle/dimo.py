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


    def invert_gate(self, k, epoch, step, b, unit):
        """Test instrument (oracle/dimo.c dimo_invert_gate): take the relu gate of (sub-net k, epoch, step, batch position b,
        hidden unit) on the other side of zero -- for pre-activations the fp64 replay shows to be at fp32 noise level."""
        lib = C.CDLL(self._lib_name)
        lib.dimo_invert_gate.argtypes = [C.c_void_p] + [C.c_int32] * 5
        if lib.dimo_invert_gate(self._h, int(k), int(epoch), int(step), int(b), int(unit)) != 0:
            raise ValueError("invert_gate: bad argument or more than 8 inversions")


def _load_general(fp64=False):
    name = os.path.join(_HERE, "libdimo_gen64.so" if fp64 else "libdimo_gen.so")
    if name not in _cache:
        if not os.path.exists(name):
            build()
        lib = C.CDLL(name)
        lib.dimog_real_bytes.restype = C.c_int
        assert lib.dimog_real_bytes() == (8 if fp64 else 4)
        _cache[name] = _cabi.bind_general_oracle(lib)
    return _cache[name]


class GeneralOracleEngine(GeneralEngine):
    """The general CPU oracle (oracle/dimo_general.c) behind the GeneralEngine face."""

    def __init__(self, D, layers, out_dim, fp64=False, **kw):
        kw.pop("device_id", None)
        super().__init__(_load_general(fp64), D, layers, out_dim, **kw)
