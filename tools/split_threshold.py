import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from ibo_amd import _lib, DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import sweep
for N, D in ((1024, 4), (2048, 8), (256, 3)):
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y, noise=.1)
    for M in (128, 512, 1024, 2048, 4096, 8192, 16384):
        cand = DeviceArray.from_host(rs.rand(M, D))
        out = []
        for path in (2, 3):
            _lib.check(_lib.lib.ibo_set_option(b"sweep_path", path))
            sweep(GP, cand)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); r = sweep(GP, cand); ts.append((time.perf_counter() - t0) * 1e6)
            out.append((np.median(ts), r["kernel_ms"] * 1e3))
        _lib.check(_lib.lib.ibo_set_option(b"sweep_path", 0))
        print("N=%d M=%5d  plain %.0f us (kernel %.0f)   split %.0f us (kernel %.0f)" % (N, M, out[0][0], out[0][1], out[1][0], out[1][1]), flush=True)
