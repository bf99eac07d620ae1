"""Where a DIRECT batch's time goes on the host side (ibo_set_option("small_trace")): staging + launch calls, the wait for the
   completion word, copying the results out.  Next to it: tools/launch_floor (what an empty launch / a resident kernel cost).
   python3 tools/time_small_split.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import maximizeEI

def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k.encode(), v))

for N, D in ((1024, 4), (2048, 8), (64, 2)):
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y)
    maximizeEI(GP, [[0., 1.]] * D)
    opt("small_trace", 1)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = maximizeEI(GP, [[0., 1.]] * D); ts.append((time.perf_counter() - t0) * 1e3)
    print("N=%4d D=%d  maximizeEI %.2f ms (min of 5)" % (N, D, min(ts)), flush=True)
    opt("small_trace", 2)
