#!/usr/bin/env python3
"""fit (device ms) in the plain single-level order and in super-panels (ibo_set_option("super_min_nb")), L and W compared bit for bit
and against NumPy (GPU box).   python3 tools/super_ab.py [N ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
sizes = [int(a) for a in sys.argv[1:]] or [2100, 3000, 3072, 3500, 4096, 5000, 6144]
D = 16
for N in sizes:
    rs = np.random.RandomState(7); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    out = {}
    for name, nb in (("plain", 1000), ("super", 32)):
        _lib.check(_lib.lib.ibo_set_option(b"super_min_nb", nb))
        g = GaussianProcess(GaussianKernel_ard([.3 * np.sqrt(D / 4.)] * D), X, Y, noise=.1)
        ms = []
        for _ in range(6):
            g._fit_device(); ms.append(g.last_fit_ms())
        W = np.empty((N, N)); _lib.check(_lib.lib.ibo_gp_get_W(g._handle(), _lib.dp(W)))
        out[name] = (float(np.median(ms[1:])), np.array(g.L), W, np.array(g.R) if name == "plain" else None)
        del g
    _lib.check(_lib.lib.ibo_set_option(b"super_min_nb", 48))
    Lr = np.linalg.cholesky(out["plain"][3])
    e = [np.abs(out[k][1] - Lr).max() for k in ("plain", "super")]
    wl = np.abs(out["super"][2][:200] @ out["super"][1] - np.eye(N)[:200]).max()
    print("N=%5d  plain %.3f ms   super %.3f ms   L equal bit for bit: %s  W: %s   |L - numpy| %.2e / %.2e   |W L - I| (200 rows) %.2e"
          % (N, out["plain"][0], out["super"][0], np.array_equal(out["plain"][1], out["super"][1]), np.array_equal(out["plain"][2], out["super"][2]),
             e[0], e[1], wl), flush=True)
