#!/usr/bin/env python3
"""the factorisation's block-column kernels on eight waves (chol_step8_kernel, chol_pipe8_kernel) against four (chol_step_kernel,
chol_pipe_kernel), and the pipelined order forced on small matrices (chol_pipe = 2): bits of L and W, device time of the fit.
python3 tools/check_step8.py [N ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard

def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k.encode(), v))
for N in [int(a) for a in sys.argv[1:]] or [200, 700, 1024, 1216, 1536, 2048, 3000, 4096]:
    D = 4 if N <= 1024 else 8
    rs = np.random.RandomState(2)
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
    res = {}
    for name, w, pipe in (("4 waves", 4, 1), ("8 waves", 8, 1), ("8 waves, fused steps", 8, 0), ("4 waves, pipelined from the first column", 4, 2)):
        opt("step_waves", w); opt("chol_pipe", pipe)
        gp = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y, noise=.1)
        dev = []
        for _ in range(9):
            gp._fit_device(); dev.append(gp.last_fit_ms())
        W = np.empty((N, N)); _lib.check(_lib.lib.ibo_gp_get_W(gp._handle(), _lib.dp(W)))
        res[name] = (gp.L.copy(), W, float(np.median(dev)), float(np.min(dev)))
    opt("step_waves", 8); opt("chol_pipe", 1)
    ref = res["4 waves"]
    print("N=%5d  " % N + "   ".join("%s %.3f ms (%s)" % (k, v[2], "same bits" if np.array_equal(v[0], ref[0]) and np.array_equal(v[1], ref[1]) else "DIFFERENT")
                                      for k, v in res.items()), flush=True)
