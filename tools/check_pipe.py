"""the software-pipelined fit (chol_pipe 1) against the step-by-step one (0): same bits in L and W; timing"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
opt = lambda k, v: _lib.check(_lib.lib.ibo_set_option(k, v))
for N in (1, 63, 64, 65, 100, 129, 500, 1000, 1024, 1500, 2048):
    rs = np.random.RandomState(N); X = rs.rand(N, 4); Y = np.sin(3 * X.sum(1))
    for ride in (1, 0):
        opt(b"chol_ride", ride)
        res = []
        for pipe in (2, 0):                      # 2: pipelined at every size
            opt(b"chol_pipe", pipe)
            GP = GaussianProcess(GaussianKernel_ard([.4] * 4), X, Y, noise=.05)
            W = np.empty((N, N)); _lib.check(_lib.lib.ibo_gp_get_W(GP._handle(), _lib.dp(W)))
            ms = []
            for _ in range(5): GP._fit_device(); ms.append(GP.last_fit_ms())
            res.append((GP.L.copy(), W, np.median(ms)))
        same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        err = np.abs(res[0][0] - np.linalg.cholesky(GP.R)).max()
        print("N=%5d ride=%d  pipe %.3f ms  steps %.3f ms  same bits %s  |L - chol(R)| %.1e" % (N, ride, res[0][2], res[1][2], same, err), flush=True)
opt(b"chol_ride", 1); opt(b"chol_pipe", 1)
