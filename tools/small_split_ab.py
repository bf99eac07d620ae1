"""one 16-candidate block per product workgroup (ibo_set_option("small_split", 1)) against both blocks in one (0): bits, then
   DIRECT and single-call latency.   python3 tools/small_split_ab.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, MaternKernel5
from ibo_amd.acquisition import maximizeEI

def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k.encode(), v))
bad = 0
for N, D in ((700, 3), (1024, 4), (2048, 8), (3000, 5)):
    rs = np.random.RandomState(N); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    for kern in (GaussianKernel_ard([.3] * D), MaternKernel5([.4, 1.0])):
        GP = GaussianProcess(kern, X, Y, noise=.01)
        for M in (1, 7, 16, 17, 33, 52, 64, 100, 250, 256):
            Q = rs.rand(M, D); out = {}
            for f in (0, 1):
                opt("small_split", f)
                mu, s2 = GP.posteriors(Q); out[f] = (np.array(mu), np.array(s2))
            if not all(np.array_equal(a, b) for a, b in zip(out[0], out[1])): bad += 1; print("DIFF", N, D, type(kern).__name__, M)
print("bit comparison: %d differences" % bad)
for f in (0, 1):
    opt("small_split", f)
    for N, D in ((1024, 4), (2048, 8), (600, 3)):
        rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
        GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y)
        maximizeEI(GP, [[0., 1.]] * D)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); r = maximizeEI(GP, [[0., 1.]] * D); ts.append((time.perf_counter() - t0) * 1e3)
        x = rs.rand(D); GP.posterior(x)
        t0 = time.perf_counter()
        for _ in range(300): GP.posterior(x)
        tp = (time.perf_counter() - t0) / 300 * 1e6
        print("small_split=%d N=%4d D=%d  maximizeEI %.2f ms (min of 7; %s)  posterior(x) %.1f us" % (f, N, D, min(ts), r[0], tp), flush=True)
opt("small_split", 1)
