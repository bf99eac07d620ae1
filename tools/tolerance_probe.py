#!/usr/bin/env python3
"""Where conditioning is worst (GPU box): N in {1000, 2048} x noise in {1e-3, 1e-4} x D in {1, 2}, clustered points,
SE and Matern-5/2.  For every case: relative errors of mu / s2 / EI of the device path against BOTH oracle flavours
(native: invR double mat-vec + libm erf; Python path: two triangular solves + NR erf), the two flavours against each
other, and all three against an 80-bit long-double reference of the same formulas (what "the right answer" is when the
two reference paths themselves disagree).
python3 tools/tolerance_probe.py [out.txt]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess import kernel as K
from ibo_amd.acquisition import sweep
from ibo_amd import _lib

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None


def say(s):
    print(s, flush=True)
    if out:
        out.write(s + "\n"); out.flush()


def clustered(rs, N, D):
    """three tight clusters plus a uniform background: near-duplicate rows in R"""
    c = rs.rand(3, D)
    parts = [c[i] + 0.02 * rs.randn(N // 4, D) for i in range(3)]
    parts.append(rs.rand(N - 3 * (N // 4), D))
    return np.clip(np.vstack(parts), 0, 1)


def truth_longdouble(R, X, Y, kfun, cand, noise):
    """mu, s2 in 80-bit arithmetic: Cholesky and solves by hand (numpy has no long-double LAPACK)"""
    ld = np.longdouble
    N = len(Y)
    A = R.astype(ld).copy()
    L = np.zeros((N, N), dtype=ld)
    for j in range(N):
        d = A[j, j] - np.dot(L[j, :j], L[j, :j])
        L[j, j] = np.sqrt(d)
        if j + 1 < N:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j].dot(L[j, :j])) / L[j, j]
    def fsolve(b):
        z = np.zeros(N, dtype=ld)
        for i in range(N):
            z[i] = (b[i] - np.dot(L[i, :i], z[:i])) / L[i, i]
        return z
    zy = fsolve(Y.astype(ld))
    mu = np.zeros(len(cand), dtype=ld); s2 = np.zeros(len(cand), dtype=ld)
    for c in range(len(cand)):
        ks = kfun(X, cand[c]).astype(ld)
        z = fsolve(ks)
        mu[c] = np.dot(z, zy)
        s2[c] = (ld(1) + ld(noise)) - np.dot(z, z)
    return mu.astype(float), s2.astype(float)


def rel(a, b, floor=1e-300):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


worst = {}
for N in (1000, 2048):
    for noise in (1e-3, 1e-4):
        for D in (1, 2):
            for kind in ("ard", "m5"):
                rs = np.random.RandomState(1000 * D + N + int(1e5 * noise))
                X = clustered(rs, N, D)
                Y = np.sin(3 * X.sum(1)) + .01 * rs.randn(N)
                hyp = np.full(D, .3) if kind == "ard" else np.r_[.5, 1.0]
                ours = K.GaussianKernel_ard(hyp) if kind == "ard" else K.MaternKernel5(hyp)
                okern = orc.Kern(kind, hyp)
                ogp = orc.GP(okern, X, Y, noise=noise)
                gp = GaussianProcess(ours, X, Y, noise=noise)
                Mo = 48
                cand = np.vstack([rs.rand(Mo // 2, D), np.clip(X[rs.randint(0, N, Mo // 2)] + 1e-3 * rs.randn(Mo // 2, D), 0, 1)])
                r_nat = sweep(gp, cand, acq='ei', xi=.01, native=True, outputs=("mu", "s2", "acq"))
                r_py = sweep(gp, cand, acq='ei', xi=.01, native=False, outputs=("mu", "s2", "acq"))
                o_nat = orc.sweep_native(ogp, cand, orc.ACQ_EI, .01)
                pm, ps = ogp.posteriors(cand)
                o_ei_py = orc.acq_value(orc.ACQ_EI, orc.ERF_NR, pm, np.sqrt(ps), np.max(Y), .01)
                kfun = lambda Xm, c: np.array([okern.cov(x, c) for x in Xm])
                t_mu, t_s2 = truth_longdouble(ogp.R, X, Y, kfun, cand, noise)
                t_s2n = np.clip(t_s2, 1e-8, 10); t_s2p = np.clip(t_s2, 1e-7, 10)
                tag = "N=%d D=%d noise=%g %s" % (N, D, noise, kind)
                row = dict(
                    ours_vs_native=(rel(r_nat["mu"], o_nat["mu"], 1e-9), rel(r_nat["s2"], o_nat["s2"]), rel(r_nat["acq"] * (np.abs(o_nat["acq"]) > 1e-12), o_nat["acq"] * (np.abs(o_nat["acq"]) > 1e-12), 1e-12)),
                    ours_vs_python=(rel(r_py["mu"], pm, 1e-9), rel(r_py["s2"], ps), rel(r_py["acq"] * (np.abs(o_ei_py) > 1e-12), o_ei_py * (np.abs(o_ei_py) > 1e-12), 1e-12)),
                    native_vs_python=(rel(o_nat["mu"], pm, 1e-9), rel(np.clip(o_nat["s2"], 1e-7, 10), ps), 0.0),
                    ours_vs_truth=(rel(r_nat["mu"], t_mu, 1e-9), rel(r_nat["s2"], t_s2n), 0.0),
                    native_vs_truth=(rel(o_nat["mu"], t_mu, 1e-9), rel(o_nat["s2"], t_s2n), 0.0),
                    python_vs_truth=(rel(pm, t_mu, 1e-9), rel(ps, t_s2p), 0.0))
                say(tag + "  min s2 %.2e" % float(np.min(t_s2)))
                for k, v in row.items():
                    say("    %-18s mu %.2e  s2 %.2e  ei %.2e" % (k, v[0], v[1], v[2]))
                    w = worst.setdefault(k, [0, 0, 0])
                    for i in range(3): w[i] = max(w[i], v[i])
say("WORST")
for k, v in worst.items():
    say("    %-18s mu %.2e  s2 %.2e  ei %.2e" % (k, v[0], v[1], v[2]))
