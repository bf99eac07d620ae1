import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
rs = np.random.RandomState(2); X = rs.rand(1024, 4); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(1024)
GP = GaussianProcess(GaussianKernel_ard([.3] * 4), X, Y, noise=.1)
ch = np.random.RandomState(102).rand(1 << 20, 4)
GP.posteriors(ch[:1000])
for pipe in (1, 0, 1):
    _lib.check(_lib.lib.ibo_set_option(b"host_pipeline", pipe))
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); GP._posterior_arrays(ch); ts.append((time.perf_counter() - t0) * 1e3)
    print("pipeline", pipe, ["%.1f" % t for t in ts])
