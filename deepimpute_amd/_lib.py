"""Loader of the product library libdimn.so (HIP/gfx950).  Fails loudly: no fallback."""
import ctypes as C
import os

from . import _cabi

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIMN_LIB_PATH: diagnostic builds of the same source (tools/ab_def.sh, tools/res_timeline.py); the product path is fixed
LIB_PATH = os.environ.get("DIMN_LIB_PATH") or os.path.join(_HERE, "csrc", "libdimn.so")
_fns = None
_lib = None


def library():
    """The loaded CDLL (for symbol checks)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "deepimpute_amd: %s is not built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (needs hipcc, --offload-arch=gfx950). There is no CPU fallback."
                % LIB_PATH)
        # multi-process GPU work (RCCL between ranks) needs dmabuf IPC on this driver stack; the variable is read when
        # the HSA runtime initialises, i.e. at the first HIP call after this load
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    return _lib


def load():
    """Bound function table of libdimn.so; raises ImportError if missing or ABI-mismatched."""
    global _fns
    if _fns is None:
        fns = _cabi.bind(library(), "dimn_", gpu=True)
        ver = fns["abi_version"]()
        if ver != _cabi.ABI_VERSION:
            raise ImportError("libdimn.so ABI version %d != expected %d; rebuild it"
                              % (ver, _cabi.ABI_VERSION))
        _fns = fns
    return _fns


def device_count():
    """HIP devices visible to this process (0 without a GPU)."""
    n = C.c_int32(0)
    load()["device_count"](C.byref(n))
    return n.value


_warm = {}


def warm_up_async(device_id=0):
    """Bring the GPU up on a helper thread (dimn_warm_up: HIP context + the pinned bounce buffers, ~0.1 s) while the caller still
    parses arguments / reads its matrix; returns at once.  Once per (process, device); silent when there is no library or GPU --
    the first real call then reports that."""
    import threading
    device_id = int(device_id)
    if device_id in _warm:
        return
    _warm[device_id] = True

    def work():
        try:
            load()["warm_up"](device_id)
        except Exception:
            pass
    threading.Thread(target=work, name="dimn-warm-up", daemon=True).start()
