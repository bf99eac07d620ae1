import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import maximizeEI
for N, D in ((1024, 4), (2048, 8), (64, 2)):
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y)
    for fp in (0, 1, 0, 1):
        _lib.check(_lib.lib.ibo_set_option(b"flag_poll", fp))
        r0 = maximizeEI(GP, [[0., 1.]] * D)
        t0 = time.perf_counter()
        for _ in range(5): r = maximizeEI(GP, [[0., 1.]] * D)
        print(N, D, "flag_poll", fp, "%.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3), r[0], flush=True)
        x = rs.rand(D)
        GP.posterior(x); t0 = time.perf_counter()
        for _ in range(200): GP.posterior(x)
        print("    posterior(x) %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
