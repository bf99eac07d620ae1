import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.run_configs import hartman6
from ibo_amd.gaussianprocess import PrefGaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
P = 512
pts = np.random.RandomState(4).rand(2 * P, 6)
prefs = []
for i in range(P):
    a, b = pts[2 * i], pts[2 * i + 1]
    prefs.append((a, b, 0) if hartman6(a) > hartman6(b) else (b, a, 0))
PrefGaussianProcess(GaussianKernel_ard([0.53, 0.57, 2.5, 0.34, 0.27, 0.35]), prefs[:8])
pr = cProfile.Profile(); pr.enable()
GP = PrefGaussianProcess(GaussianKernel_ard([0.53, 0.57, 2.5, 0.34, 0.27, 0.35]), prefs)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
