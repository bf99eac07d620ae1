"""PrefGaussianProcess.addPreferences latency (512 pairs -> 1024 points, D = 6): cold and warm, with a profile of a warm call
python3 tools/time_pref.py"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ibo_amd.gaussianprocess import PrefGaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from run_configs import hartman6
P = 512
rs = np.random.RandomState(4)
pts = rs.rand(2 * P, 6)
prefs = []
for i in range(P):
    a, b = pts[2 * i], pts[2 * i + 1]
    prefs.append((a, b, 0) if hartman6(a) > hartman6(b) else (b, a, 0))
for rep in range(4):
    t0 = time.perf_counter()
    GP = PrefGaussianProcess(GaussianKernel_ard([0.53, 0.57, 2.5, 0.34, 0.27, 0.35]), prefs)
    print("addPreferences %d: %.1f ms" % (rep, (time.perf_counter() - t0) * 1e3), flush=True)
pr = cProfile.Profile(); pr.enable()
GP = PrefGaussianProcess(GaussianKernel_ard([0.53, 0.57, 2.5, 0.34, 0.27, 0.35]), prefs)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
