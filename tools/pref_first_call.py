#!/usr/bin/env python3
"""where the first PrefGaussianProcess construction's extra milliseconds go (GPU box): every ctypes call into the library timed, first call against second"""
import sys, os, time, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ibo_amd import _lib
import bench
from ibo_amd.gaussianprocess import PrefGaussianProcess, GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
acc = collections.OrderedDict()
class Timed(object):
    def __init__(self, f, name): self.f, self.name = f, name
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.f(*a); acc.setdefault(self.name, []).append((time.perf_counter() - t0) * 1e3); return r
for name in _lib.EXPORTED:
    if name.startswith("ibo_") and name not in ("ibo_last_error",):
        setattr(_lib.lib, name, Timed(getattr(_lib.lib, name), name))
# what bench.py has done before its C4 block: a plain model and sweeps
X0 = np.random.RandomState(1).rand(1024, 4); Y0 = np.sin(X0.sum(1))
g0 = GaussianProcess(GaussianKernel_ard([.3] * 4), X0, Y0, noise=.1); del g0
rs = np.random.RandomState(4); pts = rs.rand(1024, 6); prefs = []
for i in range(512):
    a_, b_ = pts[2 * i], pts[2 * i + 1]
    prefs.append((a_, b_, 0) if bench.hartman6(a_) > bench.hartman6(b_) else (b_, a_, 0))
for call in (1, 2, 3):
    acc.clear()
    t0 = time.perf_counter()
    PG = PrefGaussianProcess(GaussianKernel_ard([0.53, 0.57, 2.5, 0.34, 0.27, 0.35]), prefs)
    tot = (time.perf_counter() - t0) * 1e3
    inlib = sum(sum(v) for v in acc.values())
    print("call %d: %.2f ms, %.2f in the library, %.2f in Python" % (call, tot, inlib, tot - inlib))
    for k, v in acc.items():
        if sum(v) > 0.05: print("    %-28s x%3d %7.2f ms  (first %.2f)" % (k, len(v), sum(v), v[0]))
    del PG
