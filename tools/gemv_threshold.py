import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from ibo_amd import DeviceArray, _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import MaternKernel5, GaussianKernel_ard
from ibo_amd.acquisition.gallery import fastUCBGallery
from ibo_amd.acquisition import maximizeEI, sweep
for N, D, kern in ((2048, 8, MaternKernel5([.5 * np.sqrt(2.), 1.0])), (1024, 4, GaussianKernel_ard([.3] * 4))):
    rs = np.random.RandomState(3)
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
    GP = GaussianProcess(kern, X, Y, noise=.1)
    for gm in (16, 8, 4, 1, 0):
        _lib.check(_lib.lib.ibo_set_option(b"gemv_max", gm))
        out = []
        for M in (1, 2, 4, 8, 12, 16, 24, 32):
            c = rs.rand(M, D)
            sweep(GP, c); 
            t0 = time.perf_counter()
            for _ in range(20): r = sweep(GP, c)
            out.append("%d:%.0f(%.0f)" % (M, (time.perf_counter() - t0) / 20 * 1e6, r["kernel_ms"] * 1e3))
        print("N=%d gemv_max=%2d  wall us (kernel us) by M: %s" % (N, gm, "  ".join(out)), flush=True)
    cand = DeviceArray.from_host(np.random.RandomState(103).rand(1 << 17, D))
    for gm in (16, 4, 0):
        _lib.check(_lib.lib.ibo_set_option(b"gemv_max", gm))
        fastUCBGallery(GP, [[0., 1.]] * D, 4, candidates=cand)
        t0 = time.perf_counter(); fastUCBGallery(GP, [[0., 1.]] * D, 8, candidates=cand); print("  gemv_max=%d gallery8 %.1f ms" % (gm, (time.perf_counter() - t0) * 1e3), flush=True)
