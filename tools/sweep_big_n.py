"""the MFMA sweep beyond 4096 observations (sweep2_kernel<.., BIGN>): python3 tools/sweep_big_n.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ibo_amd import DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import sweep
D, M = 8, 1 << 16
for N in (4096, 6000, 8192, 12000):
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    GP = GaussianProcess(GaussianKernel_ard([.5] * D), X, Y, noise=.1)
    cand = DeviceArray.from_host(rs.rand(M, D))
    for _ in range(2): r = sweep(GP, cand)
    ms = float(np.mean([sweep(GP, cand)["kernel_ms"] for _ in range(3)]))
    F = N * N + 3 * N * D + 4 * N
    print("N=%5d  fit %.1f ms  %s %.2f ms  %.1f TFLOP/s (%.0f %%)" % (N, GP.last_fit_ms(), r["kernel"], ms, F * M / ms / 1e9, F * M / ms / 1e9 / 78.6 * 100), flush=True)
