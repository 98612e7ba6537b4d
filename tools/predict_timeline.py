#!/usr/bin/env python
"""Phase timeline of k_predict: builds libdimn with -DDIMN_PRED_TL, runs the forward over all cells of BASELINE configs[2]
and prints the share of wave time per phase (s_memtime ticks of lane 0 of every wave, summed over the launch).
    python tools/predict_timeline.py [extra hipcc flags...]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "deepimpute_amd", "csrc", "libdimn_ptl.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DDIMN_PRED_TL"] + sys.argv[1:] +
                      ["-o", lib, os.path.join(ROOT, "deepimpute_amd", "csrc", "dimn.hip"), "-ldl", "-lpthread"])
os.environ["DIMN_LIB_PATH"] = lib
import time

import bench
from deepimpute_amd import _lib
from deepimpute_amd.engine import HipEngine

cfg = bench.CONFIGS["cfg3"]
norm = bench.synth_counts(cfg["n"], cfg["g"], seed=0)
targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=0)
train, val = bench.split_rows(cfg["n"], seed=0)
K = len(targets)
eng = bench.make_engine(HipEngine, cfg, targets, preds, norm, train, val, [K], [0], 0, 0, 1e-4)
eng.gather(True)
eng.init_weights()
fn = _lib.library().dimn_debug_pred_timeline
fn.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 8)()
eng.predict_device(); eng.synchronize()
fn(buf)
t0 = time.perf_counter(); eng.predict_device(); eng.synchronize(); dt = time.perf_counter() - t0
assert fn(buf) == 0
tl = np.frombuffer(buf, np.uint64).astype(np.float64)
names = ["first layer (chunk loop)", "bias/act -> LDS", "barrier", "second layer MFMAs", "softplus + stores", "tail", "-", "-"]
print("predict of %d cells x %d sub-nets: %.2f ms" % (cfg["n"], K, 1e3 * dt))
for nm, v in zip(names[:6], tl[:6]):
    print("  %-26s %5.1f %%" % (nm, 100 * v / tl[:6].sum()))
