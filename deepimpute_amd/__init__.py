"""MI355X-native MultiNet hot path of DeepImpute: `deepimpute_amd.multinet.MultiNet`, `deepimpute_amd.deepImpute.deepImpute`."""


def release_cached_memory():
    """Return the library's idle device blocks to the driver (see deepimpute_amd._lib.release_cached_memory)."""
    from . import _lib
    _lib.release_cached_memory()
