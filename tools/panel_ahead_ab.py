# The marginal-likelihood grid of BASELINE config 5 (64 theta-points, N = 4096, D = 16) with and without the panels' look-ahead
# (ibo_set_option("panel_ahead")): wall and device span per grid, and whether the values are the same bits.
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import nlml_grid
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rs = np.random.RandomState(5); X = rs.rand(N, 16); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
th = np.exp(np.random.RandomState(105).uniform(np.log(.1), np.log(3), size=(T, 16)))
vals = {}
for rep in range(2):
    for ahead in (0, 1):
        _lib.check(_lib.lib.ibo_set_option(b"panel_ahead", ahead))
        best = (1e9, 0)
        for r in range(5):
            g0 = _lib.gpu_time_ms(0); t0 = time.perf_counter()
            v = np.asarray(nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-3)[0])
            dt = (time.perf_counter() - t0) * 1e3; g = _lib.gpu_time_ms(0) - g0
            if r and dt < best[0]: best = (dt, g)
        vals[ahead] = v
        print("N=%d thetas=%d panel_ahead=%d: wall %.3f ms  device span %.3f ms" % (N, T, ahead, best[0], best[1]), flush=True)
print("same bits:", bool(np.array_equal(vals[0].view(np.int64), vals[1].view(np.int64))), " finite:", int(np.isfinite(vals[1]).sum()), "of", T)
