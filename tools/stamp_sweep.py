#!/usr/bin/env python3
"""Per-tile time breakdown of the sweep kernel from a diagnostic build with s_memrealtime stamps
(-DIBO_STAMPS, tools/libibo_hip_stamps.so).  100 MHz ticks -> microseconds.
    IBO_HIP_LIB=tools/libibo_hip_stamps.so IBO_STAMP_FILE=/tmp/st.bin python tools/stamp_sweep.py [N] [D]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import sweep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 4
M = 1 << 20
rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y)
cand = DeviceArray.from_host(np.random.RandomState(102).rand(M, D))
for _ in range(3):
    r = sweep(GP, cand)
st = np.fromfile(os.environ["IBO_STAMP_FILE"], dtype=np.uint64).reshape(-1, 16)
t = st[:, :6].astype(np.int64)
us = 0.01
names = ["candidate load + setup", "panels before the last", "last panel", "final reduce + barrier", "epilogue (wave 0)"]
d = np.diff(t, axis=1) * us
print("N=%d D=%d: kernel %.3f ms, %d tiles; per tile (median, us):" % (N, D, r["kernel_ms"], len(st)))
for i, n in enumerate(names):
    print("   %-28s %8.2f" % (n, np.median(d[:, i])))
print("   %-28s %8.2f" % ("tile total (entry->exit)", np.median((t[:, 5] - t[:, 0]) * us)))
# chain tiles per CU: same (xcc, hw_id cu bits) -> gaps between consecutive tiles
cu = (st[:, 7].astype(np.int64) << 16) | ((st[:, 6].astype(np.int64) >> 8) & 0xFFF)     # XCC + SE/SH/CU bits of HW_ID
gaps = []
for c in np.unique(cu):
    m = np.where(cu == c)[0]
    o = m[np.argsort(t[m, 0])]
    gaps.extend(((t[o[1:], 0] - t[o[:-1], 5]) * us).tolist())
gaps = np.array(gaps)
print("   distinct CU ids %d; gap exit->next entry on the same CU: median %.2f us, mean %.2f us" % (len(np.unique(cu)), np.median(gaps), gaps.mean()))
span = (t[:, 5].max() - t[:, 0].min()) * us
print("   first entry -> last exit: %.1f us; sum(tile)/256 CUs = %.1f us" % (span, (t[:, 5] - t[:, 0]).sum() * us / 256))
cy = st[:, 8:14].astype(np.float64); ns = st[:, 14:16].astype(np.float64)
for nm, o, k in (("plain stages   ", 0, 0), ("diagonal stages", 3, 1)):
    n = np.maximum(ns[:, k], 1)
    print("   wave 0, %s: %5.1f per tile; shader cycles per stage: k* gen %7.0f  MFMA loop %7.0f  barrier wait %7.0f" %
          (nm, np.median(ns[:, k]), np.median(cy[:, o] / n), np.median(cy[:, o + 1] / n), np.median(cy[:, o + 2] / n)))
print("   (one wave issues 128 MFMAs per plain stage: 4 waves/SIMD x 128 x 64 cycles = 32768 cycles if the pipe never idles)")
