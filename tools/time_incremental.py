#!/usr/bin/env python3
"""kernel and wall time of the incremental sweep (gallery rounds) at C2 / C3 / C4 sizes, for the one-kernel first sweep
(gallery_prune = 0), the two-part state with every tile completed (2) and the pruned one (1); hallucinated observations as in
fastUCBGallery.   python3 tools/time_incremental.py"""
import sys, os, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import DeviceArray, _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, MaternKernel5
from ibo_amd.acquisition import sweep

def synth(seed, N, D):
    rs = np.random.RandomState(seed); X = rs.rand(N, D)
    return X, np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
def levels(GP):
    nl = ctypes.c_int(); sp = (ctypes.c_int * 3)(); cnt = (ctypes.c_int64 * 4)()
    _lib.check(_lib.lib.ibo_sweep_state_levels(GP._handle(), ctypes.byref(nl), sp, cnt))
    return nl.value, list(sp)[:max(0, nl.value - 1)], list(cnt)[:nl.value]
def state_info(GP):
    t, c = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(_lib.lib.ibo_sweep_state_info(GP._handle(), ctypes.byref(t), ctypes.byref(c)))
    return t.value, c.value

if len(sys.argv) > 1: _lib.check(_lib.lib.ibo_set_option(b"part_levels", int(sys.argv[1])))
for N, D, kern, M in ((1024, 4, GaussianKernel_ard([.3] * 4), 1 << 20), (2048, 8, MaternKernel5([.5, 1.0]), 1 << 19),
                      (1024, 6, GaussianKernel_ard([.4] * 6), 1 << 20)):
    X, Y = synth(3, N, D)
    cand = np.random.RandomState(103).rand(M, D)
    picks = {}
    for mode in (0, 2, 1):
        _lib.check(_lib.lib.ibo_set_option(b"gallery_prune", mode))
        GP = GaussianProcess(kern, X, Y, noise=.1)
        dc = DeviceArray.from_host(cand)
        sweep(GP, dc, acq='ei', xi=.4, native=False); GP._fit_device()           # warm
        walls, info, seq = [], [], []
        lev0 = None
        for rnd in range(8):
            t0 = time.perf_counter()
            r = sweep(GP, dc, acq='ei', xi=.4, native=False, exclude=np.array(seq) if seq else None, exclude_radius=.05, incremental=True)
            walls.append((time.perf_counter() - t0) * 1e3)
            info.append(state_info(GP)[1]); seq.append(cand[r["best_idx"]])
            if rnd == 0: lev0 = levels(GP)
            GP.addData(seq[-1], GP.mu(seq[-1]))
        picks[mode] = [tuple(s) for s in seq]
        tiles = state_info(GP)[0]
        print("N=%d D=%d M=2^%d gallery_prune=%d: sweep() wall per round %s ms, total %.1f ms; complete tiles %s of %d; same picks as mode 0: %s" %
              (N, D, int(np.log2(M)), mode, " ".join("%.2f" % w for w in walls), sum(walls), info, tiles, picks[mode] == picks[0]), flush=True)
        if mode == 1:
            print("     after the first sweep: %d levels, splits at rows %s, tiles standing at each level %s" % lev0, flush=True)
_lib.check(_lib.lib.ibo_set_option(b"gallery_prune", 1))
