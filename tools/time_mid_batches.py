"""kernel time of sweep() for batches of 64..4096 candidates (small2.hip's three kernels): python3 tools/time_mid_batches.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ibo_amd import DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import sweep
for N, D in ((1024, 4), (2048, 8), (4096, 8)):
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    GP = GaussianProcess(GaussianKernel_ard([.3 * np.sqrt(D / 4.)] * D), X, Y, noise=.1)
    line = []
    for M in (64, 256, 1024, 4096):
        cand = DeviceArray.from_host(rs.rand(M, D))
        for _ in range(3): sweep(GP, cand)
        ms = np.median([sweep(GP, cand)["kernel_ms"] for _ in range(9)])
        line.append("M=%d %.0f us" % (M, ms * 1e3))
    print("N=%d  " % N + "   ".join(line), flush=True)
