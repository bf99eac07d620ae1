#!/usr/bin/env python3
"""A/B the sweep kernel's tile variants in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24).  Prints median / min kernel ms and TFLOP/s.
    python tools/ab_sweep.py [N] [D] [log2M] [variants...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import _lib, DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.acquisition import sweep

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 4
M = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 20)
variants = [int(v) for v in sys.argv[4:]] or [0, 1, 2]
rs = np.random.RandomState(2)
X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
GP = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y, noise=.1)
cand = DeviceArray.from_host(np.random.RandomState(102).rand(M, D))
F = N * N + 3 * N * D + 4 * N
res = {v: [] for v in variants}
ref = None
for rnd in range(7):
    for v in variants:
        _lib.check(_lib.lib.ibo_set_option(b"sweep_variant", v))
        r = sweep(GP, cand, acq='ei', xi=.01, native=True)
        if rnd:
            res[v].append(r["kernel_ms"])
        if ref is None:
            ref = (r["best_val"], r["best_idx"])
        assert r["best_idx"] == ref[1] and abs(r["best_val"] - ref[0]) <= 1e-12 * abs(ref[0]), (v, r, ref)
_lib.check(_lib.lib.ibo_set_option(b"sweep_variant", 2))
for v in variants:
    ms = np.array(res[v])
    print("variant %d  N=%d D=%d M=%d: median %.3f ms  min %.3f ms  -> %.2f TFLOP/s (%.1f%% of 78.6)  %.3e evals/s" %
          (v, N, D, M, np.median(ms), ms.min(), F * M / np.median(ms) / 1e9, F * M / np.median(ms) / 1e9 / 78.6 * 100,
           M / np.median(ms) * 1e3))
