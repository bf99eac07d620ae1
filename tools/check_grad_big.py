"""NLML + gradient beyond 2048 rows: the fused two-level factorisation against the unfused one (same bits) and timing.
python3 tools/check_grad_big.py"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import marginalLikelihood
for N, D in ((2300, 6), (4096, 16)):
    rs = np.random.RandomState(N); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
    k = GaussianKernel_ard([.4 * np.sqrt(D / 4.)] * D)
    out = []
    for f2 in (1, 0):
        _lib.check(_lib.lib.ibo_set_option(b"chol_fused2", f2))
        marginalLikelihood(k, X, Y, D, True, noise=1e-3)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); v, g = marginalLikelihood(k, X, Y, D, True, noise=1e-3); ts.append((time.perf_counter() - t0) * 1e3)
        out.append((v, np.array(g), np.median(ts)))
    _lib.check(_lib.lib.ibo_set_option(b"chol_fused2", 1))
    print("N=%d D=%d  fused2 %.2f ms  unfused %.2f ms   same bits: %s" % (N, D, out[0][2], out[1][2], out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])))
