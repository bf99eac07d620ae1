#!/usr/bin/env python3
"""GP-fit latency (host X,Y -> K, L, L^-1, packed W, alpha ready on the device) for a few N, and the
accuracy of the factor against numpy:  python3 tools/time_fit.py [N ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard

opts = [a for a in sys.argv[1:] if "=" in a]          # key=value: ibo_set_option before the runs
if opts:
    from ibo_amd import _lib
    for o in opts:
        k, v = o.split("="); _lib.check(_lib.lib.ibo_set_option(k.encode(), int(v)))
for N in [int(a) for a in sys.argv[1:] if "=" not in a] or [256, 1024, 2048, 4096]:
    D = 4 if N <= 1024 else 8
    rs = np.random.RandomState(2)
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
    gp = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y, noise=.1)
    host, dev = [], []
    for _ in range(7):
        t0 = time.perf_counter(); gp._fit_device(); host.append((time.perf_counter() - t0) * 1e3)
        dev.append(gp.last_fit_ms())
    L = gp.L
    Lr = np.linalg.cholesky(gp.R)
    print("N=%5d  fit host %.3f ms  device %.3f ms   max|L-L_numpy|/max|L| = %.2e" %
          (N, np.median(host), np.median(dev), np.abs(L - Lr).max() / np.abs(Lr).max()), flush=True)
