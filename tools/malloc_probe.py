#!/usr/bin/env python
"""How long does hipMalloc of a ~20 GB arena take, and when?  (round 4: fit.hand_over 0.02 / 0.19 / 0.48 / 0.62 s across runs)
Scenarios inside ONE process, timed around hipMalloc + a 4-byte memset on the block (forces the mapping):
  fresh, after-free same size, after-free larger, while the old block is still held, with host memory pressure, after touching."""
import ctypes as C
import sys
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipDeviceSynchronize.argtypes = []
hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
GB = 1 << 30


def info():
    a, b = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(a), C.byref(b))
    return "free %.1f / %.1f GB" % (a.value / GB, b.value / GB)


def malloc(gb, touch_all=False):
    p = C.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), int(gb * GB))
    t1 = time.perf_counter()
    hip.hipMemset(p, 0, int(gb * GB) if touch_all else 4)
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    print("  hipMalloc %5.2f GB rc=%d: %8.1f ms, %s %8.1f ms   [%s]" % (gb, rc, 1e3 * (t1 - t0), "memset all" if touch_all else "memset 4 B", 1e3 * (t2 - t1), info()))
    return p


def free(p):
    t0 = time.perf_counter()
    hip.hipFree(p)
    print("  hipFree: %8.1f ms   [%s]" % (1e3 * (time.perf_counter() - t0), info()))


print("fresh process", info())
a = malloc(19.5)
print("second block while the first is held")
b = malloc(19.6)
free(a)
print("same size right after the free")
a = malloc(19.5)
free(a); free(b)
print("larger than anything freed")
a = malloc(21.0)
free(a)
print("after writing the whole block once (memset all), then free, then a larger one")
a = malloc(19.5, touch_all=True)
free(a)
a = malloc(19.6)
free(a)
print("hold 60 GB in 4 GB blocks (all written), then a 19.6 GB block")
held = [malloc(4.0, touch_all=True) for _ in range(15)]
a = malloc(19.6)
free(a)
print("... free half of the 4 GB blocks, then 19.6 GB again")
for p in held[::2]:
    hip.hipFree(p)
a = malloc(19.6)
free(a)
for p in held[1::2]:
    hip.hipFree(p)
print("host memory pressure: 24 GB of touched numpy arrays, then 19.6 GB")
host = [np.ones((3 << 30) // 8) for _ in range(8)]
a = malloc(19.6)
free(a)
print("ten 19.6 GB malloc / free cycles")
for i in range(10):
    p = C.c_void_p()
    t0 = time.perf_counter(); hip.hipMalloc(C.byref(p), int(19.6 * GB)); t1 = time.perf_counter(); hip.hipFree(p); t2 = time.perf_counter()
    print("  cycle %d: malloc %.1f ms free %.1f ms" % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
