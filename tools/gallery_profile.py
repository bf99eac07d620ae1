#!/usr/bin/env python3
"""host-side profile (cProfile) of the C3 gallery of 8 on a fixed candidate array (GPU box):  python3 tools/gallery_profile.py"""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ibo_amd import DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import MaternKernel5
from ibo_amd.acquisition.gallery import fastUCBGallery
N, D, M = 2048, 8, 1 << 19
rs = np.random.RandomState(3)
X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
cand = DeviceArray.from_host(np.random.RandomState(103).rand(M, D))
mk = lambda: GaussianProcess(MaternKernel5([.4] * D), X, Y, noise=1e-3)
for _ in range(3):
    GP = mk(); t0 = time.perf_counter(); fastUCBGallery(GP, [[0., 1.]] * D, 8, candidates=cand); print("gallery %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
for _ in range(3):
    GP = mk(); pr.enable(); fastUCBGallery(GP, [[0., 1.]] * D, 8, candidates=cand); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
