#!/usr/bin/env python3
"""the diagonal block's panel on two waves (diag64_dpp.h, IBO_DIAG_SPLIT) against the one-wave panel: L, W and the NLML grid bit for bit
between two builds of the library, and the device time of a fit with each.  GPU box:
    python3 tools/check_split.py            (needs tools/libibo_hip_split.so: the library with linalg.hip compiled with -DIBO_DIAG_SPLIT=1; the shipped library is the one-wave build)"""
import os, sys, subprocess, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(HERE, ".."))
    from ibo_amd import _lib
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.gaussianprocess.trainhyper import nlml_grid
    out = {}
    for N in (60, 100, 200, 257, 700, 1024, 1500, 2048, 3000, 4096):
        D = 4 if N <= 1024 else 8
        rs = np.random.RandomState(2)
        X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
        gp = GaussianProcess(GaussianKernel_ard([.3] * D), X, Y, noise=.1)
        dev = []
        for _ in range(9):
            gp._fit_device(); dev.append(gp.last_fit_ms())
        W = np.empty((N, N)); _lib.check(_lib.lib.ibo_gp_get_W(gp._handle(), _lib.dp(W)))
        out["L%d" % N] = gp.L.copy(); out["W%d" % N] = W; out["t%d" % N] = np.median(dev)
    # a matrix that is not positive definite: the pivot reported must be the same
    Xd = np.vstack([X[:1500], X[77:78]])
    try:
        GaussianProcess(GaussianKernel_ard([.3] * 8), Xd, np.zeros(1501), noise=0.0); out["notpd"] = np.array([-1])
    except Exception as e:
        out["notpd"] = np.array([hash(str(e)) % (1 << 31)])
    rs = np.random.RandomState(5); X = rs.rand(1100, 5); Y = np.sin(3 * X.sum(1))
    thetas = np.exp(rs.uniform(np.log(.2), np.log(2.), (12, 5)))
    out["grid"] = np.asarray(nlml_grid(GaussianKernel_ard, thetas, X, Y, noise=1e-3)[0])
    np.savez(sys.argv[2], **out)
    sys.exit(0)
tmp = tempfile.mkdtemp()
res = {}
for name, lib in (("two waves", os.path.join(HERE, "libibo_hip_split.so")), ("one wave", os.path.join(HERE, "..", "ibo_amd", "libibo_hip.so"))):
    f = os.path.join(tmp, name.replace(" ", "_") + ".npz")
    env = dict(os.environ, IBO_HIP_LIB=os.path.abspath(lib))
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", f], env=env)
    res[name] = np.load(f)
a, b = res["two waves"], res["one wave"]
bad = 0
for k in a.files:
    if k.startswith("t"):
        continue
    same = np.array_equal(a[k], b[k], equal_nan=True)
    bad += not same
    if k.startswith("L"):
        N = k[1:]
        print("N=%5s  one wave %.3f ms   two waves %.3f ms   L %s, W %s" % (N, b["t" + N], a["t" + N], "identical" if same else "DIFFERENT",
              "identical" if np.array_equal(a["W" + N], b["W" + N]) else "DIFFERENT"), flush=True)
print("not-PD report identical:", np.array_equal(a["notpd"], b["notpd"]), "  NLML grid identical:", np.array_equal(a["grid"], b["grid"], equal_nan=True))
print("FAIL" if bad else "all identical")
