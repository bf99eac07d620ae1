import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
X = np.random.RandomState(1).rand(1024, 4); Y = np.sin(X.sum(1))
t0 = time.perf_counter(); g = GaussianProcess(GaussianKernel_ard([.3] * 4), X, Y, noise=.1); t1 = time.perf_counter()
print("first model %.1f ms" % ((t1 - t0) * 1e3))
for k in range(3):
    t0 = time.perf_counter(); g.posterior(np.array([.1, .2, .3, .4])); print("posterior #%d %.3f ms" % (k, (time.perf_counter() - t0) * 1e3))
g2 = GaussianProcess(GaussianKernel_ard([.3] * 4), X, Y, noise=.1)
for k in range(2):
    t0 = time.perf_counter(); g2.posterior(np.array([.1, .2, .3, .4])); print("second model posterior #%d %.3f ms" % (k, (time.perf_counter() - t0) * 1e3))
