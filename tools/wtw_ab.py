import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from ibo_amd import _lib
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood
def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k.encode(), v))
for N, D in ((1024, 4), (2048, 8), (4096, 16)):
    rs = np.random.RandomState(3)
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .05 * rs.randn(N)
    k = GaussianKernel_ard(np.full(D, .5))
    out = {}
    for w in (4, 8):
        opt("wtw_waves", w)
        marginalLikelihood(k, X, Y, D, True)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); v, g = marginalLikelihood(k, X, Y, D, True); ts.append((time.perf_counter() - t0) * 1e3)
        out[w] = (v, g, np.median(ts))
    opt("wtw_waves", 8)
    print("N=%d  4 waves %.3f ms   8 waves %.3f ms   same value and gradient: %s" % (N, out[4][2], out[8][2], out[4][0] == out[8][0] and np.array_equal(out[4][1], out[8][1])), flush=True)
