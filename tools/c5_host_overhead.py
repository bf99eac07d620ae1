import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import nlml_grid
rs = np.random.RandomState(5); X = rs.rand(4096, 16); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(4096)
th = np.exp(np.random.RandomState(105).uniform(np.log(.1), np.log(3), size=(64, 16)))
for r in range(6):
    g0 = _lib.gpu_time_ms(0); t0 = time.perf_counter()
    nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-3)
    dt = (time.perf_counter() - t0) * 1e3; g = _lib.gpu_time_ms(0) - g0
    print("wall %.3f ms  device span %.3f ms  host-only %.3f ms" % (dt, g, dt - g))
