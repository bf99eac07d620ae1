#!/usr/bin/env python3
"""gallery_first_call.py behind parts of what bench.py has run before its gallery: which of them makes the first call slower?
python3 tools/gallery_first_call_after.py [c2] [fits] [fit4096]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import MaternKernel5, GaussianKernel_ard
from ibo_amd.acquisition import sweep as _sw
import ibo_amd.acquisition.gallery as G
what = set(sys.argv[1:])
def synth(seed, N, D):
    rs = np.random.RandomState(seed); X = rs.rand(N, D); return X, np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
if "c2" in what:
    X, Y = synth(2, 1024, 4); GP2 = GaussianProcess(GaussianKernel_ard([.3] * 4), X, Y, noise=.1)
    c2 = DeviceArray.from_host(np.random.RandomState(102).rand(1 << 20, 4))
    for _ in range(5): _sw(GP2, c2, acq='ei', xi=.01, native=True)
for n, d in ((1024, 4), (2048, 8), (4096, 16)):
    if "fits" in what or (n == 4096 and "fit4096" in what):
        Xf, Yf = synth(7, n, d); g = GaussianProcess(GaussianKernel_ard([.3] * d), Xf, Yf, noise=.1)
        for _ in range(5): g._fit_device()
        del g
        g = GaussianProcess(GaussianKernel_ard([.3] * d), Xf, Yf, noise=.1, reserve_rows=8)
        for q in range(3): g.addData(np.random.RandomState(8 + q).rand(d), 0.0)
        del g
X, Y = synth(3, 2048, 8)
GP = GaussianProcess(MaternKernel5([.5, 1.0]), X, Y, noise=.1)
cand = DeviceArray.from_host(np.random.RandomState(103).rand(1 << 19, 8))
for _ in range(6): _sw(GP, cand, acq='ei', xi=.01, native=True)
acc = {}
def timed(name, f):
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return r
    return g
G._start = timed("start", G._start); G.maximizeEI = timed("DIRECT", G.maximizeEI); G.sweep = timed("sweep", G.sweep)
_add = GaussianProcess.addData; GaussianProcess.addData = timed("addData", _add)
for call in (1, 2):
    acc.clear()
    t0 = time.perf_counter(); G.fastUCBGallery(GP, [[0., 1.]] * 8, 8, candidates=cand); tot = (time.perf_counter() - t0) * 1e3
    print("%s call %d: %.2f ms  " % (sorted(what), call, tot) + "  ".join("%s %.2f = %s" % (k, sum(v), " ".join("%.2f" % x for x in v)) for k, v in acc.items()))
