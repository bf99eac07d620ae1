#!/usr/bin/env python3
"""two steps per pass in the pipelined factorisation (chol_pipe8_kernel<1> / <2>, "pipe_pairs") against one step per pass: L and W bit for
bit, and the device time of a fit either way.  GPU box:  python3 tools/check_pairs.py [N ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard

def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k.encode(), v))
bad = 0
for N in [int(a) for a in sys.argv[1:]] or [300, 700, 1024, 1100, 1536, 2048, 2304, 2370, 2560, 3000, 3500, 4096, 5000]:
    D = 4 if N <= 1024 else 8
    rs = np.random.RandomState(2)
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
    res = {}
    for name, pairs in (("one step per pass", 0), ("two steps per pass", 1)):
        opt("pipe_pairs", pairs)
        gp = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y, noise=.1)
        dev = []
        for _ in range(9):
            gp._fit_device(); dev.append(gp.last_fit_ms())
        W = np.empty((N, N)); _lib.check(_lib.lib.ibo_gp_get_W(gp._handle(), _lib.dp(W)))
        res[name] = (gp.L.copy(), W, float(np.median(dev)), float(np.min(dev)))
    opt("pipe_pairs", 36)
    a, b = res["one step per pass"], res["two steps per pass"]
    ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    bad += not ok
    print("N=%5d  one step per pass %.3f ms   two steps per pass %.3f ms   L and W %s" % (N, a[2], b[2], "identical" if ok else "DIFFERENT"), flush=True)
print("FAIL" if bad else "all identical")
