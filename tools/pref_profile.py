import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, ".")
from ibo_amd.gaussianprocess import PrefGaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
sys.path.insert(0, "tools")
from run_configs import hartman6
P = 512
rs = np.random.RandomState(4)
pts = rs.rand(2 * P, 6)
prefs = []
for i in range(P):
    a, b = pts[2 * i], pts[2 * i + 1]
    prefs.append((a, b, 0) if hartman6(a) > hartman6(b) else (b, a, 0))
k = lambda: GaussianKernel_ard([0.53, 0.57, 2.5, 0.34, 0.27, 0.35])
for _ in range(4):
    t0 = time.perf_counter(); PrefGaussianProcess(k(), prefs); print("addPreferences %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): PrefGaussianProcess(k(), prefs)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
