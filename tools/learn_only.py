#!/usr/bin/env python3
"""N evaluations of NLML + gradient (ibo_nlml_grad) at one size, for the profiler:  python3 tools/learn_only.py 4096 16 [count]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood
N, D = int(sys.argv[1]), int(sys.argv[2])
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rs = np.random.RandomState(9); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
k = GaussianKernel_ard([.5] * D)
ms = []
for _ in range(cnt):
    t0 = time.perf_counter(); marginalLikelihood(k, X, Y, D, True, noise=1e-3); ms.append((time.perf_counter() - t0) * 1e3)
print("N=%d D=%d  %d evaluations, median %.3f ms" % (N, D, cnt, float(np.median(ms[1:]))))
