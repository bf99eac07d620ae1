import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import maximizeEI
for N, D in ((1024, 4), (2048, 8), (64, 2)):
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y)
    maximizeEI(GP, [[0., 1.]] * D)
    t0 = time.perf_counter(); maximizeEI(GP, [[0., 1.]] * D); print(N, D, (time.perf_counter() - t0) * 1e3, "ms")
