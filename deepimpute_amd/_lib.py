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


def _join_warm_ups():
    """atexit: a warm-up thread still inside hipHostMalloc / hipMalloc when the interpreter tears down can hang or crash the
    process (a short script, a failed argument check, inspect_data() calling exit(1)); wait for it first."""
    for t in list(_warm.values()):
        if t is not True and t.is_alive():
            t.join(timeout=30.0)


def warm_up_async(device_id=0):
    """Bring the GPU up on a helper thread (dimn_warm_up: HIP context + the pinned bounce buffers, ~0.1 s) while the caller still
    parses arguments / reads its matrix; returns at once.  Once per (process, device); silent when there is no library or GPU --
    the first real call then reports that.  Called by what is about to use the GPU anyway (the top of MultiNet.fit / predict, the
    CLI before it parses its CSV, bench.py at start-up) -- never by a constructor: building a MultiNet has no side effects, as in
    the reference, so load-only use, inspection, or a fork() after construction see no HIP context."""
    import atexit
    import threading
    device_id = int(device_id)
    if device_id in _warm:
        return
    if not _warm:
        atexit.register(_join_warm_ups)

    def work():
        try:
            load()["warm_up"](device_id)
        except Exception:
            pass
    t = threading.Thread(target=work, name="dimn-warm-up", daemon=True)
    _warm[device_id] = t
    t.start()


def release_cached_memory():
    """Give the process-wide cache of large device blocks (include/dimn.h: dimn_release_cached_memory) back to the driver: after
    MultiNet.close() / engine.close() the blocks of >= 32 MB otherwise wait, up to DIMN_ARENA_CACHE_GB (default 48), for the next
    fit() of this process.  MultiNet.close() calls it.  A no-op when the library was never loaded."""
    if _fns is not None:
        _fns["release_cached_memory"]()


def cached_memory_info():
    """(bytes of idle device blocks in the library's cache, bytes of cached-class blocks handles currently own)."""
    out = (C.c_int64 * 2)()
    load()["cached_memory_info"](out)
    return int(out[0]), int(out[1])
