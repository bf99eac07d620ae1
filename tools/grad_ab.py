#!/usr/bin/env python3
"""the round-4 gradient kernel (nlml_grad_fast_kernel, "grad_ard") against the first one: value identical, gradient to rounding; time of
one NLML + gradient evaluation either way.  GPU box:  python3 tools/grad_ab.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ibo_amd import _lib
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, MaternKernel5, SVGaussianKernel_iso
from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood


def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k.encode(), v))


for N, D in ((100, 2), (257, 3), (1024, 4), (1500, 11), (2048, 8), (3000, 32), (4096, 16)):
    rs = np.random.RandomState(3)
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .05 * rs.randn(N)
    for k, nh, nm in ((GaussianKernel_ard(np.full(D, .5) + .1 * rs.rand(D)), D, "SE-ARD"), (MaternKernel5([.8, 1.1]), 2, "Matern-5/2"),
                      (SVGaussianKernel_iso([.7, 1.2]), 2, "SE-iso + signal")):
        out = {}
        for fast in (0, 1):
            opt("grad_ard", fast)
            marginalLikelihood(k, X, Y, nh, True)
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); v, g = marginalLikelihood(k, X, Y, nh, True); ts.append((time.perf_counter() - t0) * 1e3)
            out[fast] = (v, np.atleast_1d(np.array(g)), np.median(ts))
        opt("grad_ard", 1)
        err = np.abs(out[0][1] - out[1][1]).max() / np.abs(out[0][1]).max()
        print("N=%5d D=%2d %-16s first kernel %.3f ms   round-4 kernel %.3f ms   value identical: %s   gradient: max |difference| / max |g| = %.1e" %
              (N, D, nm, out[0][2], out[1][2], out[0][0] == out[1][0], err), flush=True)
