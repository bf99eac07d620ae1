#!/usr/bin/env python3
"""Per-tile time of sweep2_kernel from the diagnostic build (make -C ibo_amd/csrc stamps): entry -> prologue done -> panels done ->
exit, and the gap between a tile's exit and the next tile's entry on the same CU (100 MHz s_memrealtime ticks).
    IBO_HIP_LIB=tools/libibo_hip_stamps.so IBO_STAMP_FILE=/tmp/s2.bin python3 tools/stamp_sweep2.py [N] [D]"""
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
st = np.fromfile(os.environ["IBO_STAMP_FILE"], dtype=np.uint64).reshape(-1, 8)
t = st[:, :4].astype(np.int64); us = 0.01
d = np.diff(t, axis=1) * us
print("N=%d D=%d: kernel %.3f ms (diagnostic build), %d tiles; per tile (median / mean, us):" % (N, D, r["kernel_ms"], len(st)))
for i, n in enumerate(["prologue (table, alpha, candidates)", "panels", "final reduce + store"]):
    print("   %-38s %8.2f %8.2f" % (n, np.median(d[:, i]), d[:, i].mean()))
cu = (st[:, 5].astype(np.int64) << 16) | ((st[:, 4].astype(np.int64) >> 8) & 0xFFF)
gaps = []
for c in np.unique(cu):
    m = np.where(cu == c)[0]; o = m[np.argsort(t[m, 0])]
    gaps.extend(((t[o[1:], 0] - t[o[:-1], 3]) * us).tolist())
gaps = np.array(gaps)
span = (t[:, 3].max() - t[:, 0].min()) * us
busy = (t[:, 3] - t[:, 0]).sum() * us / len(np.unique(cu))
print("   CUs seen %d; exit -> next entry on the same CU: median %.2f us, mean %.2f us" % (len(np.unique(cu)), np.median(gaps), gaps.mean()))
print("   first entry -> last exit %.1f us; mean busy time per CU %.1f us (%.1f %%); last CU idle at the end for %.1f us on average" %
      (span, busy, 100 * busy / span, np.mean([t[:, 3].max() - t[cu == c, 3].max() for c in np.unique(cu)]) * us))
