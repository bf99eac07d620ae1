import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import _lib
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import nlml_grid
rs = np.random.RandomState(5); X = rs.rand(4096, 16); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(4096)
th = np.exp(np.random.RandomState(105).uniform(np.log(.1), np.log(3), size=(64, 16)))
if len(sys.argv) > 1: _lib.check(_lib.lib.ibo_set_option(b"chol_update2", int(sys.argv[1])))
for _ in range(3):
    t0 = time.perf_counter(); nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-3); print("%.2f ms" % ((time.perf_counter() - t0) * 1e3))
