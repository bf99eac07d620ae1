import os, sys
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/ibo_amd") else ".")
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, MaternKernel5, GaussianKernel_iso
def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k.encode(), v))
for N, D in ((100, 2), (257, 3), (700, 4), (1024, 4), (1500, 8), (2048, 8), (2500, 16), (3000, 5), (1000, 40)):
    rs = np.random.RandomState(N)
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1))
    out = []
    for kern in (GaussianKernel_ard([.3] * D), MaternKernel5([.4] * D) if D <= 32 else GaussianKernel_iso([.5])):
        res = []
        for cf in (0, 1):
            opt("cov_fit", cf)
            gp = GaussianProcess(kern, X, Y, noise=1e-3)
            dev = []
            for _ in range(7):
                gp._fit_device(); dev.append(gp.last_fit_ms())
            W = np.empty((N, N)); _lib.check(_lib.lib.ibo_gp_get_W(gp._handle(), _lib.dp(W)))
            res.append((gp.L.copy(), W, np.array(gp.R), float(np.median(dev))))
        opt("cov_fit", 1)
        same = all(np.array_equal(res[0][i], res[1][i]) for i in range(3))
        out.append("%s %.3f -> %.3f ms %s" % (type(kern).__name__[:10], res[0][3], res[1][3], "same bits" if same else "DIFFERENT"))
    print("N=%5d D=%2d  " % (N, D) + "   ".join(out), flush=True)
