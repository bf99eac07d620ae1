"""chol_panel_rows_kernel against the launch sequence it replaces: NLML grid values must be identical bit for bit;
C5 timing both ways.   python3 tools/check_panel_rows.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import _lib
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import nlml_grid
for N, D, T in ((700, 3, 5), (1500, 6, 7), (4096, 16, 64)):
    rs = np.random.RandomState(N); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    th = np.exp(rs.uniform(np.log(.1), np.log(3), size=(T, D)))
    out = {}
    for flag in (0, 1):
        _lib.check(_lib.lib.ibo_set_option(b"chol_panel_rows", flag))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); v = np.asarray(nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-3)[0]); best = min(best, time.perf_counter() - t0)
        out[flag] = v
        print("N=%d  panel_rows=%d  %.2f ms  (%.3f ms/theta)" % (N, flag, best * 1e3, best * 1e3 / T), flush=True)
    print("   identical:", np.array_equal(out[0], out[1]), " max rel diff %.2e" % np.max(np.abs(out[0] - out[1]) / np.abs(out[0])))
