import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import EI
for N, D in ((32, 2), (1024, 4), (2048, 8)):
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y, noise=.1)
    x = rs.rand(D); GP.posterior(x)
    t0 = time.perf_counter()
    for _ in range(200): GP.posterior(x)
    tp = (time.perf_counter() - t0) / 200 * 1e6
    ei = EI(GP, .01); ei.negf(x)
    t0 = time.perf_counter()
    for _ in range(200): ei.negf(x)
    te = (time.perf_counter() - t0) / 200 * 1e6
    t0 = time.perf_counter()
    for _ in range(20): GP.addData(rs.rand(D), 0.1)
    ta = (time.perf_counter() - t0) / 20 * 1e3
    print("N=%4d D=%d  posterior(x) %.0f us   EI.negf(x) %.0f us   addData (1 point) %.2f ms" % (N, D, tp, te, ta))
