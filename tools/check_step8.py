#!/usr/bin/env python3
"""fused block-column steps on eight waves (chol_step8_kernel) against four (chol_step_kernel): bits of L and W, device time.
python3 tools/check_step8.py [N ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard

def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k.encode(), v))
for N in [int(a) for a in sys.argv[1:]] or [200, 700, 1024, 1216, 2048, 3000, 4096]:
    D = 4 if N <= 1024 else 8
    rs = np.random.RandomState(2)
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
    res = {}
    for w in (4, 8):
        opt("step_waves", w)
        if N <= 1280: opt("chol_pipe", 0)
        gp = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y, noise=.1)
        dev = []
        for _ in range(9):
            gp._fit_device(); dev.append(gp.last_fit_ms())
        W = np.empty((N, N)); _lib.check(_lib.lib.ibo_gp_get_W(gp._handle(), _lib.dp(W)))
        res[w] = (gp.L.copy(), W, float(np.median(dev)), float(np.min(dev)))
        opt("chol_pipe", 1)
    opt("step_waves", 8)
    same = np.array_equal(res[4][0], res[8][0]) and np.array_equal(res[4][1], res[8][1])
    print("N=%5d  4 waves %.3f ms (min %.3f)   8 waves %.3f ms (min %.3f)   L and W identical: %s" %
          (N, res[4][2], res[4][3], res[8][2], res[8][3], same), flush=True)
