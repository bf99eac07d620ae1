#!/usr/bin/env python3
"""GPU-side timeline of DIRECT's small batches from the diagnostic build (make -C ibo_amd/csrc stamps -> tools/libibo_hip_stamps.so):
s_memrealtime stamps (100 MHz) at entry / operands in / compute done / exit of every workgroup of the three kernels.
    IBO_HIP_LIB=tools/libibo_hip_stamps.so IBO_STAMP_FILE=/tmp/ss.bin python3 tools/stamp_small.py [N] [D]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import maximizeEI
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 4
path = os.environ["IBO_STAMP_FILE"]
rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y)
maximizeEI(GP, [[0., 1.]] * D)
if os.path.exists(path): os.remove(path)
maximizeEI(GP, [[0., 1.]] * D)
rec = np.fromfile(path, dtype=np.uint64).reshape(-1, 8 + 3 * 1024 * 4)
rows = []
for r in rec:
    M, Npad, ctiles, nst, nrb, nfin = [int(v) for v in r[:6]]
    st = r[8:].astype(np.int64).reshape(3, 1024, 4)
    k = st[0, :ctiles * nst]; w = st[1, :min(1024, ctiles * nrb)]; f = st[2, :nfin]
    t0 = k[:, 0].min()
    us = lambda x: (x - t0) * 0.01
    rows.append([M, us(k[:, 0].max()), us(k[:, 1].max()), us(k[:, 2].max()), us(k[:, 3].max()),
                 us(w[:, 0].min()), us(w[:, 0].max()), us(w[:, 1].max()), us(w[:, 2].max()), us(w[:, 3].max()),
                 us(f[:, 0].min()), us(f[:, 1].max()), us(f[:, 2].max()), us(f[:, 3].max())])
rows = np.array(rows)
names = ["candidates in the batch", "k*: last workgroup enters", "k*: exponents (MFMA) done", "k*: values stored", "k*: last exit",
         "W k*: first workgroup enters", "W k*: last enters", "W k*: products done (last)", "W k*: past the barrier", "W k*: last exit",
         "finish: enters", "finish: partial sums in", "finish: values out", "finish: flag written / exit"]
print("N=%d D=%d, %d batches of one maximizeEI; microseconds after the first k* workgroup's entry (median over batches):" % (N, D, len(rows)))
for i, n in enumerate(names):
    print("   %-34s %8.2f" % (n, np.median(rows[:, i])))
