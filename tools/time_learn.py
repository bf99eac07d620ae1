#!/usr/bin/env python3
"""cost of one NLML + gradient evaluation (ibo_nlml_grad) and of a BFGS hyper-parameter fit (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scipy.optimize import fmin_bfgs
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood, nlml, dnlml

for N, D in ((256, 3), (1024, 4), (2048, 8), (4096, 16)):
    rs = np.random.RandomState(3)
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .05 * rs.randn(N)
    k = GaussianKernel_ard(np.full(D, .5))
    marginalLikelihood(k, X, Y, D, True)
    t0 = time.perf_counter()
    for _ in range(5): v, g = marginalLikelihood(k, X, Y, D, True)
    per = (time.perf_counter() - t0) / 5 * 1e3
    line = "N=%4d D=%2d  NLML+gradient %.2f ms" % (N, D, per)
    if N <= 2048:
        t0 = time.perf_counter()
        th, fopt, gopt, Bopt, fc, gc, warn = fmin_bfgs(nlml, np.log(np.full(D, .5)), dnlml, args=(GaussianKernel_ard, X, Y), maxiter=30,
                                                       disp=False, full_output=True)
        line += "   BFGS (<=30 it, %d evaluations: its line search is sensitive to the last bits) %.0f ms -> theta %s" % (
            fc, (time.perf_counter() - t0) * 1e3, np.round(np.exp(th), 3))
    print(line, flush=True)
