import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ibo_amd import DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import MaternKernel5
from ibo_amd.acquisition.gallery import fastUCBGallery
rs = np.random.RandomState(3); X = rs.rand(2048, 8); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(2048)
GP = GaussianProcess(MaternKernel5([.5, 1.0]), X, Y, noise=.1)
cand = DeviceArray.from_host(np.random.RandomState(103).rand(1 << 19, 8))
fastUCBGallery(GP, [[0., 1.]] * 8, 2, candidates=cand, seed=1)
pr = cProfile.Profile(); pr.enable()
g = fastUCBGallery(GP, [[0., 1.]] * 8, 8, candidates=cand, seed=1)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
