#!/usr/bin/env python
"""Absolute-time trace of the register-resident epoch kernel (dimn_resident.h, -DDIMN_RES_TL2): thread 0 of every workgroup stamps the
chip-wide 100 MHz clock (s_memrealtime, 10 ns) at 14 points of four chosen optimiser steps; outside those steps a mark is one compare,
so the steady state is the shipped one.  Prints, per role, the median time of every mark relative to the start of the sub-net's step
(the earliest loop top of its workgroups), i.e. the critical path across workgroups in one time base, and the step period.
    python tools/res_trace.py [K=5] [extra hipcc flags...]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = os.path.join(ROOT, "deepimpute_amd", "csrc", "libdimn_tl2.so")
if not os.environ.get("RES_TRACE_NO_BUILD"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-march=x86-64-v3", "-DDIMN_RES_TL2"] + sys.argv[2:] +
                          ["-o", lib, os.path.join(ROOT, "deepimpute_amd", "csrc", "dimn.hip"), "-ldl", "-lpthread"])
os.environ["DIMN_LIB_PATH"] = lib
import bench
from deepimpute_amd import _lib
from deepimpute_amd.engine import HipEngine

cfg = bench.CONFIGS["cfg3"]
norm = bench.synth_counts(cfg["n"], cfg["g"], seed=0)
targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=0)
train, val = bench.split_rows(cfg["n"], seed=0)
eng = bench.make_engine(HipEngine, cfg, targets[:K], preds[:K], norm, train, val, [K], [0], 0, 0, 1e-4)
eng.gather(True)
eng.init_weights()
eng.set_profiling(True)
for e in range(3):
    eng.train_epoch(e)
tm = eng.get_timers()
print("resident steps %d, %.2f us per step (HIP events)" % (tm[7], 1e3 * tm[6] / max(1, tm[7])))
G = 256 // K // 16 * 16
n = K * G
buf = (C.c_ulonglong * (n * 128))()
fn = _lib.library().dimn_debug_res_trace
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, n * 128) == 0
tl = np.frombuffer(buf, np.uint64).reshape(K, G, 4, 32).astype(np.float64) * 0.01          # us
names = ["loop top", "M1: siblings' P seen", "M1: Dd tile stored", "A: own Dd tiles seen (wave 0)", "A: Z partial -> LDS", "A: barrier (Z partials)",
         "A: barrier (dZ tile)", "A: dD partials stored", "A: gW2 + Adam done", "dD / dA seen", "dA stored (manager) / read (sibling)", "tile loop starts",
         "tile loop ends", "P published", "own half of dA stored", "-"] + ["tile loop ends, wave %d" % w for w in range(8)] + ["tile loop starts, wave %d" % w for w in range(8)]
S1 = G // 16
wi = np.arange(G)
sp = wi // 16
roles = [("role 2 + sibling (wi < 32, sp < S1 - 2)", (wi < 32) & (sp < S1 - 2)), ("role 2 + co-manager (sp = S1 - 2)", (wi < 32) & (sp == S1 - 2)),
         ("manager (last split)", sp == S1 - 1), ("sibling only", (wi >= 32) & (sp < S1 - 1))]
period = np.median(tl[:, :, 1:, 0] - tl[:, :, :-1, 0])
print("step period (loop top to loop top, median over workgroups and steps): %.2f us" % period)
t0 = tl[:, :, :, 0].min(axis=1, keepdims=True)                 # start of the step of each (sub-net, step): its earliest loop top
for label, sel in roles:
    if not sel.any():
        continue
    print(label)
    for i, nm in enumerate(names):
        v = tl[..., i][:, sel, :]
        ok = v > 0
        if not ok.any():
            continue
        rel = (v - np.broadcast_to(t0, tl[:, :, :, 0].shape)[:, sel, :])[ok]
        print("  %-40s median %7.2f us   min %7.2f   max %7.2f" % (nm, np.median(rel), rel.min(), rel.max()))

# per hidden tile: when its manager stores the Dd tile / its dA tile, when its workgroups' tile loops end (skew between the sixteen chains of a sub-net)
print("per hidden tile (median over sub-nets and steps, us after the sub-net's step start):  Dd stored | dA stored | tile loop ends (manager) | P published (siblings, max)")
rel = tl - t0[..., None]
for ht in range(16):
    m = (S1 - 1) * 16 + ht
    sib = [sp_ * 16 + ht for sp_ in range(S1 - 1)]
    print("  ht %2d  %6.2f  %6.2f  %6.2f  %6.2f" % (ht, np.median(rel[:, m, :, 2]), np.median(rel[:, m, :, 10]), np.median(rel[:, m, :, 12]),
                                             np.median(rel[:, sib, :, 13].max(axis=1)) if sib else 0.0))
