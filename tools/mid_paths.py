"""batches of 4097..16384 candidates: the panel-split kernel (sweep_path 3) against the large-batch kernel (2) and the default"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ibo_amd import _lib, DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import sweep
for N, D in ((1024, 4), (2048, 8), (256, 3)):
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    GP = GaussianProcess(GaussianKernel_ard([.3 * np.sqrt(D / 4.)] * D), X, Y, noise=.1)
    for M in (2048, 4096, 4097, 6000, 8192, 8193, 12000, 16384):
        cand = DeviceArray.from_host(rs.rand(M, D))
        line = []
        for path in (0, 2, 3):
            _lib.check(_lib.lib.ibo_set_option(b"sweep_path", path))
            for _ in range(2): r = sweep(GP, cand)
            ms = np.median([sweep(GP, cand)["kernel_ms"] for _ in range(7)])
            line.append("%s %.0f us (%s)" % (("auto", "", "tile", "split")[path], ms * 1e3, r["kernel"][:14]))
        print("N=%d M=%5d  " % (N, M) + "   ".join(line), flush=True)
_lib.check(_lib.lib.ibo_set_option(b"sweep_path", 0))
