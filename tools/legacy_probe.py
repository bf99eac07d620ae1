#!/usr/bin/env python3
"""legacy acqmaxGP (invR handed in) against the reference's compiled libego on clustered, badly conditioned data:
per-point differences over many probes, and where a DIRECT run forks.   python3 tools/legacy_probe.py [noise] [legacy_exact 1|0]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ibo_amd import _lib
from oracle import oracle

noise = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
ref = oracle.RefLib()
libc = ctypes.CDLL(None); libc.free.argtypes = [ctypes.c_void_p]
f64, dp = _lib.f64, _lib.dp
N, D = 1000, 2
exact = int(sys.argv[2]) if len(sys.argv) > 2 else 1
_lib.check(_lib.lib.ibo_set_option(b"legacy_exact", exact))
print("legacy_exact = %d" % exact)
for kind, hyp in (("ard", [.3, .3]), ("m3", [.5, 1.0])):
    rs = np.random.RandomState(77)
    c = rs.rand(3, D)
    X = np.clip(np.vstack([c[i] + 0.02 * rs.randn(N // 4, D) for i in range(3)] + [rs.rand(N - 3 * (N // 4), D)]), 0, 1)
    Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
    ogp = oracle.GP(oracle.Kern(kind, hyp), X, Y, noise=noise)
    invR = f64(np.linalg.inv(ogp.R))
    Xc, Yc, hy = f64(X), f64(Y), f64(ogp.kern.c_hyper)
    z = np.zeros(1)

    def ours(lb, ub, acq, parm, maxiter):
        lb, ub = f64(lb), f64(ub)
        r = _lib.lib.acqmaxGP(D, dp(lb), dp(ub), dp(invR), dp(Xc), dp(Yc), N, acq, int(ogp.kern.ktype), dp(hy), 0, dp(z), dp(z), 0.0,
                              dp(z), dp(z), float(parm), float(noise), maxiter, 30, 10000)
        res = np.array([r[i] for i in range(D + 1)]); libc.free(r)
        return -res[0], res[1:]
    probes = np.vstack([rs.rand(150, D), np.clip(X[rs.randint(0, N, 150)] + 1e-3 * rs.randn(150, D), 0, 1)])
    sw = oracle.sweep_native(ogp, probes, oracle.ACQ_EI, .01, invR=invR)
    for acq, parm, name in ((oracle.ACQ_EI, .01, "EI"), (oracle.ACQ_UCB, 1.3, "UCB")):
        d = []
        for x in probes:
            a, _ = ours(x, x, acq, parm, 0)
            b, _ = ref.acqmax(ogp, [[v, v] for v in x], acq, parm, maxiter=0, invR=invR)
            d.append((abs(a - b) / max(abs(b), 1e-12), a, b))
        d = np.array(d)
        k = int(np.argmax(d[:, 0]))
        print("%s %s noise %g: per-point worst rel diff %.3e (ours %.12g, libego %.12g, s2 there %.3e), median %.2e, > 1e-6: %d of %d" %
              (kind, name, noise, d[k, 0], d[k, 1], d[k, 2], sw["s2"][k], np.median(d[:, 0]), int(np.sum(d[:, 0] > 1e-6)), len(d)))
        for it in range(0, 11):
            a, ax = ours([0.] * D, [1.] * D, acq, parm, it)
            b, bx = ref.acqmax(ogp, [[0., 1.]] * D, acq, parm, maxiter=it, invR=invR)
            o, ox, ns = oracle.acqmax_native(ogp, [[0., 1.]] * D, acq, parm, maxiter=it, invR=invR)
            print("   maxiter %2d: ours %.12g at %s | libego %.12g at %s | oracle %.12g (%d samples) | rel %.2e" %
                  (it, a, np.round(ax, 6), b, np.round(bx, 6), o, ns, abs(a - b) / max(abs(b), 1e-300)))
