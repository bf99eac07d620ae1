#!/usr/bin/env python3
"""maximizeEI (DIRECT, default budget) with the resident evaluation server and by launches: wall time and the two results, which must agree
bit for bit (GPU box).   python3 tools/direct_server_ab.py [N,D ...]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, MaternKernel5
from ibo_amd.acquisition import maximizeEI
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(64, 2), (500, 3), (1024, 4), (2048, 8)]
for N, D in shapes:
    rs = np.random.RandomState(2); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    kern = MaternKernel5([.5, 1.0]) if D == 8 else GaussianKernel_ard([.3] * D)
    GP = GaussianProcess(kern, X, Y, noise=.1)
    res = {}
    for mode in (1, 0, 1, 0):
        _lib.check(_lib.lib.ibo_set_option(b"direct_resident", mode))
        maximizeEI(GP, [[0., 1.]] * D)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); r = maximizeEI(GP, [[0., 1.]] * D); ts.append((time.perf_counter() - t0) * 1e3)
        res.setdefault(mode, []).append((np.median(ts), min(ts), r))
    _lib.check(_lib.lib.ibo_set_option(b"direct_resident", 1))
    a, b = res[1][-1], res[0][-1]
    same = a[2][0] == b[2][0] and np.array_equal(np.asarray(a[2][1]), np.asarray(b[2][1]))
    print("N=%5d D=%d  resident %.3f ms (min %.3f; first round %.3f)   launches %.3f ms (min %.3f)   same result bit for bit: %s   opt %.17g"
          % (N, D, a[0], a[1], res[1][0][0], b[0], b[1], same, a[2][0]), flush=True)
