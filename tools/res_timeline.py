#!/usr/bin/env python
"""Phase timeline of the register-resident epoch kernel (dimn_resident.h): builds libdimn with -DDIMN_RES_TL, runs
one rank's share of the 8-GPU job (bench.py --limit-subnets 5) for a few epochs and prints, per role, the mean
shader-clock time per optimiser step spent in each phase (thread 0 of every workgroup).
    python tools/res_timeline.py [K=5] [extra hipcc flags...]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = os.path.join(ROOT, "deepimpute_amd", "csrc", "libdimn_tl.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DDIMN_RES_TL"] + sys.argv[2:] +
                      ["-o", lib, os.path.join(ROOT, "deepimpute_amd", "csrc", "dimn.hip"), "-ldl"])
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
steps = -(-train.size // cfg["B"])
print("resident steps %d, %.2f us per step (HIP events)" % (tm[7], 1e3 * tm[6] / max(1, tm[7])))
G = 256 // K // 16 * 16
n = K * G
buf = (C.c_ulonglong * (n * 16))()
fn = _lib.library().dimn_debug_res_timeline
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, n * 16) == 0
tl = np.frombuffer(buf, np.uint64).reshape(n, 16).astype(np.float64) / steps
names = ["M1 P gather + Dd tile", "A wait Dd", "A Dd tiles -> LDS", "A Z+loss", "A dD^T + publish", "pre-wait (X requests)", "wait dD / dA", "M2 dD sum + dA / dA tile",
         "tile loop", "P reduce + publish", "loop top", "A gW2 + Adam", "-", "-", "-", "-"]
S1 = G // 16
wi = np.arange(n) % G
sp = wi // 16
groups = [("role 2 + sibling (wi < 32, not manager)", (wi < 32) & (sp < S1 - 1)), ("manager (last split)", sp == S1 - 1),
          ("sibling only", (wi >= 32) & (sp < S1 - 1))]
for label, sel in groups:
    if not sel.any():
        continue
    print(label)
    for i, nm in enumerate(names[:12]):
        v = tl[sel, i]
        print("  %-28s mean %8.0f clk  min %8.0f  max %8.0f" % (nm, v.mean(), v.min(), v.max()))
    print("  total %.0f clk per step" % tl[sel, :12].sum(axis=1).mean())
