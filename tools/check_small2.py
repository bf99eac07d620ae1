#!/usr/bin/env python3
"""small-batch sweep (small2.hip) against the panel-split kernel and the GEMV kernel; maximizeEI latency with either"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, GaussianKernel_iso, MaternKernel3, MaternKernel5
from ibo_amd.acquisition import sweep, maximizeEI

def synth(seed, N, D):
    rs = np.random.RandomState(seed); X = rs.rand(N, D)
    return X, np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
def opt(k, v): _lib.check(_lib.lib.ibo_set_option(k, v))

ok = True
for N, D, kern, M in [(1024, 4, GaussianKernel_ard([.3] * 4), 57), (200, 3, GaussianKernel_iso([.4]), 17), (64, 1, GaussianKernel_iso([.4]), 33),
                      (1000, 6, MaternKernel3([.6, 1.0]), 600), (2048, 8, MaternKernel5([.5, 1.0]), 31), (1500, 5, GaussianKernel_ard([.3] * 5), 4096),
                      (700, 10, GaussianKernel_ard([.5] * 10), 100), (1100, 16, GaussianKernel_ard([.9] * 16), 333), (4096, 16, GaussianKernel_ard([.9] * 16), 64)]:
    X, Y = synth(N + D, N, D)
    GP = GaussianProcess(kern, X, Y, noise=.1)
    cand = np.random.RandomState(N).rand(M, D); cand[min(M - 1, 7)] = X[5]
    opt(b"small2", 0); r0 = sweep(GP, cand, outputs=("mu", "s2", "acq"))
    opt(b"small2", 1); r1 = sweep(GP, cand, outputs=("mu", "s2", "acq"))
    opt(b"sweep_path", 1); rg = sweep(GP, cand[:40], outputs=("mu", "s2", "acq")); opt(b"sweep_path", 0)
    assert r1["kernel"] == "wk_small_kernel", r1["kernel"]
    e = [np.max(np.abs(r1[k] - r0[k]) / (np.abs(r0[k]) + 1e-9)) for k in ("mu", "s2", "acq")]
    eg = [np.max(np.abs(r1[k][:40] - rg[k]) / (np.abs(rg[k]) + 1e-9)) for k in ("mu", "s2", "acq")]
    good = max(e) < 1e-8 and max(eg) < 1e-8 and r0["best_idx"] == r1["best_idx"]
    ok &= good
    print("N=%5d D=%2d M=%5d  vs split %.1e %.1e %.1e  vs gemv %.1e %.1e %.1e  argmax %d/%d  kernel %.1f us (split %.1f us) %s" %
          (N, D, M, e[0], e[1], e[2], eg[0], eg[1], eg[2], r0["best_idx"], r1["best_idx"], r1["kernel_ms"] * 1e3, r0["kernel_ms"] * 1e3, "ok" if good else "MISMATCH"), flush=True)
print("ALL OK" if ok else "FAILURES")
for N, D in ((1024, 4), (2048, 8), (64, 2)):
    X, Y = synth(2, N, D)
    GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y)
    res = {}
    for name, s2, zc in (("old", 0, 0), ("small2", 1, 0), ("small2+zero-copy", 1, 1), ("old+zero-copy", 0, 1)):
        opt(b"small2", s2); opt(b"zero_copy", zc)
        maximizeEI(GP, [[0., 1.]] * D)
        t0 = time.perf_counter()
        for _ in range(3): r = maximizeEI(GP, [[0., 1.]] * D, return_samples=True)
        res[name] = ((time.perf_counter() - t0) / 3 * 1e3, r)
    opt(b"small2", 1); opt(b"zero_copy", 1)
    same = all(abs(res[k][1][0] - res["old"][1][0]) < 1e-9 * abs(res["old"][1][0]) + 1e-14 and np.array_equal(res[k][1][1], res["old"][1][1])
               and res[k][1][2] == res["old"][1][2] for k in res)
    print("maximizeEI N=%d D=%d: " % (N, D) + "  ".join("%s %.2f ms" % (k, v[0]) for k, v in res.items()) + "   same optimum: %s" % same, flush=True)
