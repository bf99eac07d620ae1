#!/usr/bin/env python3
"""packed-panel trailing update (update2.hip) against the 64x64-tile kernel: bit-identical NLML values and factors,
and the time of the C5 grid with either."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import nlml_grid

def synth(seed, N, D):
    rs = np.random.RandomState(seed); X = rs.rand(N, D)
    return X, np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)

def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k, v))

ok = True
for N, D, T in ((700, 5, 3), (1500, 8, 5), (4096, 16, 4), (2500, 3, 2)):
    X, Y = synth(5, N, D)
    th = np.exp(np.random.RandomState(105).uniform(np.log(.1), np.log(3), size=(T, D)))
    opt(b"chol_update2", 0); v0, _ = nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-3)
    opt(b"chol_update2", 1); v1, _ = nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-3)
    same = np.array_equal(v0, v1)
    ok &= same
    print("N=%d D=%d: nlml grid identical: %s  %s" % (N, D, same, v1[:2]))
# the fit path above N = 2048 (single matrix, two-level order)
X, Y = synth(7, 3000, 6)
opt(b"chol_update2", 0); L0 = GaussianProcess(GaussianKernel_ard([.3] * 6), X, Y, noise=.1).L.copy()
opt(b"chol_update2", 1); g = GaussianProcess(GaussianKernel_ard([.3] * 6), X, Y, noise=.1); L1 = g.L
same = np.array_equal(np.tril(L0), np.tril(L1)); ok &= same
print("N=3000 fit: factor identical:", same, " fit", g.last_fit_ms(), "ms")
print("ALL OK" if ok else "FAILURES")
X, Y = synth(5, 4096, 16)
th = np.exp(np.random.RandomState(105).uniform(np.log(.1), np.log(3), size=(64, 16)))
for v in (0, 1, 0, 1):
    opt(b"chol_update2", v)
    nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-3)
    t0 = time.perf_counter(); nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-3); dt = time.perf_counter() - t0
    print("C5 64 theta, update2=%d: %.1f ms total, %.3f ms/theta, %.1f TFLOP/s" % (v, dt * 1e3, dt * 1e3 / 64, 2.336e10 * 64 / dt / 1e12), flush=True)
for N in (4096,):
    X, Y = synth(7, N, 16)
    for v in (0, 1):
        opt(b"chol_update2", v)
        g = GaussianProcess(GaussianKernel_ard([.5] * 16), X, Y, noise=.1); g._fit_device()
        print("fit N=%d update2=%d: %.3f ms" % (N, v, g.last_fit_ms()))
