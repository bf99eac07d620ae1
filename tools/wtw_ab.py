import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from ibo_amd import _lib
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood
def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k.encode(), v))
for N, D in ((1024, 4), (1536, 6), (2048, 8), (3000, 8), (4096, 16)):
    rs = np.random.RandomState(3)
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .05 * rs.randn(N)
    k = GaussianKernel_ard(np.full(D, .5))
    out = {}
    for w, x in ((4, 0), (8, 0), (8, 32)):
        opt("wtw_waves", w); opt("wtw_xcd", x)
        marginalLikelihood(k, X, Y, D, True)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); v, g = marginalLikelihood(k, X, Y, D, True); ts.append((time.perf_counter() - t0) * 1e3)
        out[(w, x)] = (v, g, np.median(ts))
    opt("wtw_waves", 8); opt("wtw_xcd", 32)
    same = all(out[k][0] == out[(4, 0)][0] and np.array_equal(out[k][1], out[(4, 0)][1]) for k in out)
    print("N=%d  4 waves %.3f ms   8 waves %.3f ms   8 waves, tiles dealt to the XCDs in 8 x 8 super-blocks %.3f ms   same value and gradient: %s" % (N, out[(4, 0)][2], out[(8, 0)][2], out[(8, 32)][2], same), flush=True)
