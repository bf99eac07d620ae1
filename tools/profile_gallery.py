"""where a gallery of 8 spends its time (C3 shard: N = 2048, D = 8, Matern-5/2, 2^19 candidates): python3 tools/profile_gallery.py"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ibo_amd import DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import MaternKernel5
from ibo_amd.acquisition.gallery import fastUCBGallery
rs = np.random.RandomState(3); N, D = 2048, 8
X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
GP = GaussianProcess(MaternKernel5([.5 * np.sqrt(2.), 1.0]), X, Y, noise=.1)
cand = DeviceArray.from_host(np.random.RandomState(103).rand(1 << 19, D))
for _ in range(2):
    t0 = time.perf_counter(); fastUCBGallery(GP, [[0., 1.]] * D, 8, candidates=cand); print("gallery %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable()
fastUCBGallery(GP, [[0., 1.]] * D, 8, candidates=cand)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
