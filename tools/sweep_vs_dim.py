import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from ibo_amd import DeviceArray, _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, MaternKernel5
from ibo_amd.acquisition import sweep
# python tools/sweep_vs_dim.py [variant [D ...]]   variant 4: sweep2_kernel (default), 2: the first-generation tile kernel
N, M = 1024, 1 << 18
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dims = [int(v) for v in sys.argv[2:]] or [2, 4, 5, 8, 9, 12, 16, 17, 20, 24, 28, 32]
_lib.check(_lib.lib.ibo_set_option(b"sweep_variant", variant))
for D in dims:
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    for name, k in (("SE", GaussianKernel_ard([.3 * np.sqrt(D / 4.)] * D)), ("M5", MaternKernel5([.5 * np.sqrt(D / 4.), 1.0]))):
        GP = GaussianProcess(k, X, Y, noise=.1)
        cand = DeviceArray.from_host(rs.rand(M, D))
        for _ in range(3):
            sweep(GP, cand)                                   # clocks and caches settle
        ms = float(np.mean([sweep(GP, cand)["kernel_ms"] for _ in range(5)]))
        F = N * N + 3 * N * D + 4 * N
        print("D=%2d %s  kernel %.2f ms  %.1f TFLOP/s (%.0f %%)" % (D, name, ms, F * M / ms / 1e9, F * M / ms / 1e9 / 78.6 * 100), flush=True)
