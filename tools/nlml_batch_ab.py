"""NLML grid, matrices per batched factorisation: python3 tools/nlml_batch_ab.py N D n_theta B [B ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import nlml_grid
N, D, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rs = np.random.RandomState(5); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
th = np.exp(np.random.RandomState(105).uniform(np.log(.1), np.log(3), size=(T, D)))
ref = None
for B in [int(b) for b in sys.argv[4:]]:
    _lib.check(_lib.lib.ibo_set_option(b"nlml_batch", B))
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); v = np.array(nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-3)[0]); ts.append((time.perf_counter() - t0) * 1e3)
    if ref is None: ref = v
    print("N=%d D=%d %d theta  batch %4d: %.2f ms = %.1f us/theta   same values %s" % (N, D, T, B, min(ts[1:]), min(ts[1:]) / T * 1e3, np.array_equal(v, ref)), flush=True)
_lib.check(_lib.lib.ibo_set_option(b"nlml_batch", 0))
