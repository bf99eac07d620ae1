#!/usr/bin/env python3
"""one maximizeEI (DIRECT, 50 iterations) at three model sizes; key=value arguments are ibo_set_option switches, e.g. sweep_path=3 (GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import maximizeEI
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("="); _lib.check(_lib.lib.ibo_set_option(k.encode(), int(v)))
for N, D in ((1024, 4), (2048, 8), (64, 2)):
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y)
    maximizeEI(GP, [[0., 1.]] * D)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); r = maximizeEI(GP, [[0., 1.]] * D); ts.append((time.perf_counter() - t0) * 1e3)
    print(N, D, "%.3f ms (median of 7, min %.3f)" % (np.median(ts), min(ts)), " optimum", r[0] if isinstance(r, tuple) else r)
