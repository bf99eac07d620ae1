#!/usr/bin/env python3
"""Randomised check of ibo_nlml_grid (GPU box): random N, D, kernel family, number of theta-points and noise -- the left-looking
grid against the right-looking one (identical values expected) and against the oracle's plain-C NLML.
python3 tools/fuzz_nlml.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import oracle as orc
from ibo_amd import _lib
from ibo_amd.gaussianprocess import kernel as K
from ibo_amd.gaussianprocess.trainhyper import nlml_values

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
special = [1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 319, 320, 383, 384, 385, 447, 448, 511, 512, 513, 640, 767, 768, 1023, 1024, 1025, 1280, 1536, 2047, 2048]
worst = 0.0; bad = 0
t0 = time.time()
for case in range(ncases):
    N = int(special[case] if case < len(special) else rs.randint(1, 1800))
    D = int(rs.randint(1, 17))
    kind = ["ard", "iso", "m3", "m5"][rs.randint(4)]
    nth = int([1, 2, 3, 7, 8, 9, 16, 17, 33][rs.randint(9)])
    noise = float([.1, .01, 1e-3][rs.randint(3)])
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
    kerns, okerns = [], []
    for t in range(nth):
        th = np.exp(rs.uniform(np.log(.2), np.log(2.), size=D)) * max(1.0, np.sqrt(D / 8.))
        if kind == "ard": kerns.append(K.GaussianKernel_ard(th)); okerns.append(orc.Kern("ard", th))
        elif kind == "iso": kerns.append(K.GaussianKernel_iso(th[:1])); okerns.append(orc.Kern("iso", th[:1]))
        elif kind == "m3": kerns.append(K.MaternKernel3(np.r_[th[0], 1.0])); okerns.append(orc.Kern("m3", np.r_[th[0], 1.0]))
        else: kerns.append(K.MaternKernel5(np.r_[th[0], 1.0])); okerns.append(orc.Kern("m5", np.r_[th[0], 1.0]))
    vals = {}
    for left in (1, 0):
        _lib.check(_lib.lib.ibo_set_option(b"chol_left", left))
        vals[left] = nlml_values(kerns, X, Y, noise)
    _lib.check(_lib.lib.ibo_set_option(b"chol_left", 1))
    same = np.array_equal(vals[1], vals[0], equal_nan=True)
    nchk = min(nth, 3 if N > 600 else nth)
    o = np.array([orc.nlml_c(okerns[t], X, Y, noise) for t in range(nchk)])
    fin = np.isfinite(o) & np.isfinite(vals[1][:nchk])
    err = float(np.max(np.abs(vals[1][:nchk][fin] - o[fin]) / np.maximum(np.abs(o[fin]), 1.0))) if fin.any() else 0.0
    nanmatch = np.array_equal(np.isfinite(o), np.isfinite(vals[1][:nchk]))
    worst = max(worst, err)
    ok = same and err < 1e-8 and nanmatch
    bad += not ok
    print("N=%5d D=%2d %-3s theta=%2d noise=%g  left==right %s  rel err vs oracle %.1e  nan pattern %s%s" % (N, D, kind, nth, noise, same, err, nanmatch, "" if ok else "   <-- FAIL"), flush=True)
print("worst relative error %.2e over %d cases, %d failures, %.1f s" % (worst, ncases, bad, time.time() - t0))
