#!/usr/bin/env python3
"""Where the first fastUCBGallery call's extra milliseconds go (GPU box): the C3 shard's model and 2^19 candidates, the call's phases timed
on the host for the FIRST call in the process and for the second -- model start (a new handle + fit), the eight DIRECT runs, the eight sweeps
(the first one forms the kept state), the hallucinated addData.   python3 tools/gallery_first_call.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import MaternKernel5
import ibo_amd.acquisition.gallery as G

rs = np.random.RandomState(3)
X = rs.rand(2048, 8); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(2048)
GP = GaussianProcess(MaternKernel5([.5, 1.0]), X, Y, noise=.1)
cand = DeviceArray.from_host(np.random.RandomState(103).rand(1 << 19, 8))
from ibo_amd.acquisition import sweep as _sw
for _ in range(3): _sw(GP, cand, acq='ei', xi=.01, native=True)          # what bench.py has run before its gallery: the full sweep kernel

acc = {}
def timed(name, f):
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return r
    return g
G._start = timed("start (new handle, buffers, fit of 2048 rows)", G._start)
G.maximizeEI = timed("DIRECT", G.maximizeEI)
G.sweep = timed("sweep", G.sweep)
_add = GaussianProcess.addData
GaussianProcess.addData = timed("addData (hallucinated point)", _add)
for call in (1, 2, 3):
    acc.clear()
    t0 = time.perf_counter(); G.fastUCBGallery(GP, [[0., 1.]] * 8, 8, candidates=cand); tot = (time.perf_counter() - t0) * 1e3
    print("call %d: %.2f ms" % (call, tot))
    for k, v in acc.items():
        print("    %-48s %7.2f ms  = %s" % (k, sum(v), " ".join("%.2f" % x for x in v)))
