#!/usr/bin/env python3
"""kernel and wall time of the incremental sweep (gallery rounds) at C2 / C3 sizes"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, MaternKernel5
from ibo_amd.acquisition import sweep, maximizeEI

def synth(seed, N, D):
    rs = np.random.RandomState(seed); X = rs.rand(N, D)
    return X, np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)

for N, D, kern, M in ((1024, 4, GaussianKernel_ard([.3] * 4), 1 << 20), (2048, 8, MaternKernel5([.5, 1.0]), 1 << 19)):
    X, Y = synth(3, N + 8, D)
    GP = GaussianProcess(kern, X[:N], Y[:N], noise=.1)
    dc = DeviceArray.from_host(np.random.RandomState(103).rand(M, D))
    for rnd in range(5):
        t0 = time.perf_counter()
        r = sweep(GP, dc, acq='ei', xi=.4, native=False, exclude=X[:rnd + 1], incremental=True)
        t1 = time.perf_counter()
        GP.addData(X[N + rnd], Y[N + rnd])
        t2 = time.perf_counter()
        o, ox = maximizeEI(GP, [[0., 1.]] * D, xi=.3)
        t3 = time.perf_counter()
        print("N=%d round %d: %-20s kernel %.3f ms, sweep() wall %.3f ms, addData %.3f ms, maximizeEI %.2f ms" %
              (N, rnd, r["kernel"], r["kernel_ms"], (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
