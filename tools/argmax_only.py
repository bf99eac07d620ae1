#!/usr/bin/env python3
"""the exact arg-max WITHOUT per-candidate values (kept-state first sweep on a new array) at a config's shape: wall time, levels, tiles per level.
python3 tools/argmax_only.py [c2|c3|c4|n4096|n3000] [part_levels] [repeats]"""
import sys, os, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import DeviceArray, _lib
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, MaternKernel5
from ibo_amd.acquisition import sweep

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
if len(sys.argv) > 2: _lib.check(_lib.lib.ibo_set_option(b"part_levels", int(sys.argv[2])))
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, D, kern, M, kw = {"c2": (1024, 4, GaussianKernel_ard([.3] * 4), 1 << 20, dict(acq='ei', xi=.01, native=True)),
                     "c3": (2048, 8, MaternKernel5([.5, 1.0]), 1 << 19, dict(acq='ei', xi=.3, native=True)),
                     "c4": (1024, 6, GaussianKernel_ard([.4] * 6), 1 << 20, dict(acq='ei', xi=.4, native=False)),
                     "n4096": (4096, 16, GaussianKernel_ard([.6] * 16), 1 << 17, dict(acq='ei', xi=.01, native=True)),
                     "n3000": (3000, 5, MaternKernel5([.4, 1.0]), 1 << 18, dict(acq='ucb', native=True))}[cfg]
rs = np.random.RandomState(3); X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
cand = np.random.RandomState(103).rand(M, D)
GP = GaussianProcess(kern, X, Y, noise=.1)
full = sweep(GP, DeviceArray.from_host(cand), **kw)
ms = []
for _ in range(reps):
    ca = DeviceArray.from_host(cand)
    _lib.check(_lib.lib.ibo_device_synchronize(0))
    t0 = time.perf_counter()
    r = sweep(GP, ca, incremental=True, **kw)
    ms.append((time.perf_counter() - t0) * 1e3)
    nl = ctypes.c_int(); sp = (ctypes.c_int * 3)(); cnt = (ctypes.c_int64 * 4)()
    _lib.check(_lib.lib.ibo_sweep_state_levels(GP._handle(), ctypes.byref(nl), sp, cnt))
    del ca
print("%s N=%d D=%d M=2^%d: arg-max only %.2f ms (min %.2f) = %.3g candidates/s; full sweep kernel %.2f ms; same index %s, value rel diff %.1e; %d levels, splits %s, tiles at each level %s" %
      (cfg, N, D, int(np.log2(M)), np.median(ms[1:]), min(ms), M / np.median(ms[1:]) * 1e3, full["kernel_ms"], r["best_idx"] == full["best_idx"],
       abs(r["best_val"] - full["best_val"]) / abs(full["best_val"]), nl.value, list(sp)[:nl.value - 1], list(cnt)[:nl.value]))
