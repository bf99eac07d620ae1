#!/usr/bin/env python3
"""
Generate the golden vectors in tests/golden/*.npz from the REAL reference.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite never execute this -- they read the committed .npz files.  The
reference is Python 2 + C++; following SURVEY.md 8(c) this script makes a
scratch copy OUTSIDE the repo, converts it with lib2to3, applies the five
mechanical NumPy-2/Python-3 fixes, compiles cpp/ with g++ -std=gnu++98 and
imports the result.  Nothing of the reference's text is stored in the
fixtures: they hold inputs and numeric outputs only.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import ctypes
import ctypes.util
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = os.environ.get("IBO_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def prepare_reference():
    tmp = tempfile.mkdtemp(prefix="ibo_ref_")
    py = os.path.join(tmp, "py"); cpp = os.path.join(tmp, "cpp")
    os.makedirs(py); os.makedirs(cpp)
    shutil.copytree(os.path.join(REF, "ego"), os.path.join(py, "ego"))
    shutil.copy(os.path.join(REF, "demo.py"), py)
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", "ego", "demo.py"], cwd=py,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def sub(path, a, b):
        p = os.path.join(py, path)
        s = open(p).read().replace(a, b)
        open(p, "w").write(s)
    sub("ego/gaussianprocess/__init__.py", "copy=False", "copy=None")
    sub("ego/randomforest/__init__.py", "copy=False", "copy=None")
    sub("ego/acquisition/__init__.py", "subplot, poly_between, draw", "subplot, draw")
    sub("ego/acquisition/__init__.py", "NA/2", "NA//2")
    # config-3 only: MaternKernel5.cov prints instead of returning (kernel.py:249)
    sub("ego/gaussianprocess/kernel.py", "        print(z)\n", "        return z\n")
    for f in ("direct", "optimizeGP", "optimizeRF", "helpers"):
        subprocess.check_call(["g++", "-std=gnu++98", "-O2", "-fPIC", "-w", "-I" + os.path.join(REF, "cpp"),
                               "-c", os.path.join(REF, "cpp", f + ".cpp"), "-o", os.path.join(cpp, f + ".o")])
    so = os.path.join(cpp, "libego.so")
    subprocess.check_call(["g++", "-shared", "-o", so] + [os.path.join(cpp, f + ".o")
                                                         for f in ("direct", "optimizeGP", "optimizeRF", "helpers")])
    orig = ctypes.util.find_library

    def fl(name):
        if name == "ego":
            return so
        if name == "libc":
            return orig("c")
        return orig(name)
    ctypes.util.find_library = fl
    os.environ["MPLBACKEND"] = "Agg"
    sys.path.insert(0, py)
    return tmp, so


TMP, LIBEGO = prepare_reference()

from ego.gaussianprocess import GaussianProcess, PrefGaussianProcess, CDF, PDF          # noqa: E402
from ego.gaussianprocess.kernel import (GaussianKernel_ard, GaussianKernel_iso, MaternKernel3,   # noqa: E402
                                        MaternKernel5, SVGaussianKernel_iso, SVGaussianKernel_ard)
from ego.gaussianprocess.prior import RBFNMeanPrior                                       # noqa: E402
from ego.gaussianprocess.trainhyper import marginalLikelihood                             # noqa: E402
from ego.acquisition import maximizeEI, maximizePI, maximizeUCB, EI, PI, UCB              # noqa: E402
import ego.acquisition.gallery as refgallery                                               # noqa: E402
from ego.utils.latinhypercube import lhcSample                                            # noqa: E402
from ego.utils.optimize import direct, cdirect                                            # noqa: E402
from ego.utils.testfunctions import Shekel5, Branin, Hartman6, Hartman3                   # noqa: E402

_DP = ctypes.POINTER(ctypes.c_double)
_lib = ctypes.CDLL(LIBEGO)
_lib.acqmaxGP.restype = _DP
_lib.acqmaxGP.argtypes = [ctypes.c_int, _DP, _DP, _DP, _DP, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_int, _DP,
                          ctypes.c_int, _DP, _DP, ctypes.c_double, _DP, _DP, ctypes.c_double, ctypes.c_double,
                          ctypes.c_int, ctypes.c_int, ctypes.c_int]


def ktype_of(k):
    if isinstance(k, GaussianKernel_ard): return 0
    if isinstance(k, GaussianKernel_iso): return 1
    if isinstance(k, MaternKernel3): return 2
    return 3


def native_point_values(GP, Q, acq, parm):
    """Per-candidate value of libego's negei/negpi/negucb: acqmaxGP with every
    dimension fixed (lb == ub == q) and maxiter=0 evaluates the objective
    exactly once, at q (cpp/direct.cpp:116-117,355)."""
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    dp = lambda a: a.ctypes.data_as(_DP)
    if isinstance(GP, PrefGaussianProcess) and GP.C is not None:
        invR = f64(np.linalg.inv(GP.R + np.linalg.inv(GP.C)))
    else:
        invR = f64(np.linalg.inv(GP.R))
    X = f64(GP.X); Y = f64(GP.Y); hyp = f64(GP.kernel.hyperparams)
    if GP.prior is None:
        nb = 0; pm = pb = pl = pw = np.zeros(1); pt = 0.0
    else:
        nb = len(GP.prior.means); pm = f64(np.array(GP.prior.means).reshape(-1)); pb = f64(GP.prior.beta)
        pl = f64(GP.prior.lowerb); pw = f64(GP.prior.width); pt = float(GP.prior.theta)
    out = np.empty(len(Q))
    D = X.shape[1]
    for i, q in enumerate(Q):
        q = f64(q)
        r = _lib.acqmaxGP(D, dp(q), dp(q), dp(invR), dp(X), dp(Y), len(Y), acq, ktype_of(GP.kernel), dp(hyp),
                          nb, dp(pm), dp(pb), pt, dp(pl), dp(pw), float(parm), float(GP.noise), 0, 30, 10)
        out[i] = -r[0]
    return out


def py_point_values(GP, Q, xi_ei=.01, xi_pi=.01, NA=None):
    ei = EI(GP, xi=xi_ei); pi = PI(GP, xi=xi_pi); ucb = UCB(GP, NA if NA else GP.X.shape[1])
    return (np.array([-ei.negf(q) for q in Q]), np.array([-pi.negf(q) for q in Q]),
            np.array([-ucb.negf(q) for q in Q]))


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)
    print("wrote", name, {k: np.shape(v) for k, v in kw.items()})


# ---------------------------------------------------------------- G1 demo
def g1_demo():
    kernel = GaussianKernel_ard(np.array([.5, .5, .3]))
    GP = GaussianProcess(kernel, noise=.1)
    X = [np.array([1, 1.5, 0.9]), np.array([.8, -.2, -0.1]), np.array([2, .8, -.2]), np.array([0, 0, .5])]
    Y = [1, .7, .6, -.1]
    GP.addData(X, Y)
    tq = np.array([[1, 1.45, 1.0], [-10, .5, -10]])
    post = np.array([GP.posterior(q) for q in tq])
    bound = [[1, 1], [-1.99, 1.98], [-1.99, 1.98]]
    opt, optx = maximizeEI(GP, bound, xi=.1)
    popt, poptx = maximizeEI(GP, bound, xi=.1, useCDIRECT=False)
    save("g1_demo", X=np.array(X, float), Y=np.array(Y, float), hyper=[.5, .5, .3], noise=.1,
         R=GP.R, L=GP.L, probe=tq, post=post, bounds=np.array(bound, float), xi=.1,
         c_opt=opt, c_optx=optx, py_opt=popt, py_optx=poptx)


# ---------------------------------------------------------------- G2 MATLAB known answers
def g2_hyper():
    X = [np.array([.5, .1, .3]), np.array([.9, 1.2, .1]), np.array([.55, .234, .1]), np.array([.234, .547, .675])]
    Y = np.array([.5, 1., .5, 2.])
    out = dict(X=np.array(X), Y=Y)
    for name, k, nh in (("ard", GaussianKernel_ard(np.array([2., 2., .1])), 3),
                        ("sviso", SVGaussianKernel_iso(np.array([1.5, 1.1])), 2),
                        ("m3", MaternKernel3(np.array([1.5, 1.1])), 2),
                        ("m5", MaternKernel5(np.array([1.5, 1.1])), 2)):
        for noise in (0.0, 1e-3):
            v, g = marginalLikelihood(k, X, Y, nh, computeGradient=True, useCholesky=True, noise=noise)
            tag = "%s_n%d" % (name, 0 if noise == 0 else 1)
            out[tag + "_nlml"] = v; out[tag + "_grad"] = g
        out[name + "_hyper"] = np.array(k.hyperparams)
        for h in range(nh):
            out["%s_d%d" % (name, h)] = k.derivative(X, h)
    save("g2_hyper", **out)


# ---------------------------------------------------------------- G3 seeded maximise* cases
def g3_cases():
    cases = []
    f1 = lambda x: float(np.sin(x * 5.))
    X = lhcSample([[0., 1.]], 5, seed=22)
    cases.append(("s22_ard1d", GaussianKernel_ard(np.array([1.0])), X, [f1(x) for x in X], .1, [[0., 1.]], None))
    f2 = lambda x: np.sum(np.sin(x))
    b2 = [[0., 5.], [0., 5.]]
    X = lhcSample(b2, 5, seed=23)
    cases.append(("s23_iso2d", GaussianKernel_iso(np.array([1.0])), X, [f2(x) for x in X], .1, b2, None))
    X = lhcSample(b2, 5, seed=24)
    cases.append(("s24_ard2d", GaussianKernel_ard(np.array([1.0, 1.0])), X, [f2(x) for x in X], .1, b2, None))
    S5 = Shekel5()
    X = lhcSample(S5.bounds, 10, seed=0)
    Ys = [S5.f(x) for x in X]
    cases.append(("s0_shekel_iso2", GaussianKernel_iso([.2]), X, Ys, .1, S5.bounds, None))
    cases.append(("s0_shekel_iso3", GaussianKernel_iso([.3]), X, Ys, .1, S5.bounds, None))
    tf = Branin()
    X = lhcSample(tf.bounds, 10, seed=0)
    Yb = [tf.f(x) for x in X]
    for nz, tag in ((1e-4, "n1e4"), (0.01, "n1e2"), (0.1, "n1e1")):
        cases.append(("s0_branin_m3_" + tag, MaternKernel3([1.0, 1.0]), X, Yb, nz, tf.bounds, None))
    pX = lhcSample(S5.bounds, 100, seed=511)
    pY = [S5.f(x) for x in pX]
    prior = RBFNMeanPrior()
    prior.train(pX, pY, bounds=S5.bounds, k=10, seed=504)
    X = lhcSample(S5.bounds, 10, seed=512)
    cases.append(("s512_prior_ard", GaussianKernel_ard([.1] * 4), X, [S5.f(x) for x in X], .1, S5.bounds, prior))

    out = {}
    names = []
    for name, kernel, X, Y, noise, bounds, prior in cases:
        GP = GaussianProcess(kernel, X, Y, prior=prior, noise=noise)
        D = GP.X.shape[1]
        probe = np.array(lhcSample(bounds, 16, seed=900 + len(names)))
        post = np.array([GP.posterior(q) for q in probe])
        ei_py, pi_py, ucb_py = py_point_values(GP, probe, NA=D)
        t = len(GP.Y) + 1
        ucb_parm = np.sqrt(0.2 * 2.0 * np.log(t ** (D // 2 + 2) * np.pi ** 2 / (3.0 * 0.1)))
        ei_c = native_point_values(GP, probe, 0, .01)
        pi_c = native_point_values(GP, probe, 1, .01)
        ucb_c = native_point_values(GP, probe, 2, ucb_parm)
        eif = EI(GP); pif = PI(GP)
        mi = 10
        d_ei = direct(eif.negf, bounds, maxiter=mi)
        c_ei = cdirect(eif.negf, bounds, maxiter=mi)
        m_ei = maximizeEI(GP, bounds, maxiter=mi)
        d_pi = direct(pif.negf, bounds, maxiter=mi)
        m_pi = maximizePI(GP, bounds, maxiter=mi)
        m_ucb = maximizeUCB(GP, bounds, maxiter=mi)
        m_ei50 = maximizeEI(GP, bounds)
        p = name + "/"
        out.update({p + "X": GP.X, p + "Y": GP.Y, p + "noise": noise, p + "bounds": np.array(bounds, float),
                    p + "ktype": ktype_of(kernel), p + "hyper": np.array(kernel.hyperparams, float),
                    p + "R": GP.R, p + "L": GP.L, p + "probe": probe, p + "post": post,
                    p + "ei_py": ei_py, p + "pi_py": pi_py, p + "ucb_py": ucb_py,
                    p + "ei_c": ei_c, p + "pi_c": pi_c, p + "ucb_c": ucb_c, p + "ucb_parm": ucb_parm,
                    p + "direct_ei": np.r_[d_ei[0], d_ei[1]], p + "cdirect_ei": np.r_[c_ei[0], c_ei[1]],
                    p + "max_ei": np.r_[m_ei[0], m_ei[1]], p + "direct_pi": np.r_[d_pi[0], d_pi[1]],
                    p + "max_pi": np.r_[m_pi[0], m_pi[1]], p + "max_ucb": np.r_[m_ucb[0], m_ucb[1]],
                    p + "max_ei50": np.r_[m_ei50[0], m_ei50[1]]})
        if prior is not None:
            out.update({p + "pmeans": np.array(prior.means), p + "pbeta": np.array(prior.beta),
                        p + "ptheta": prior.theta, p + "plowerb": np.array(prior.lowerb),
                        p + "pwidth": np.array(prior.width),
                        p + "prior_mu": np.array([prior.mu(q) for q in probe])})
        names.append(name)
    out["names"] = np.array(names)
    save("g3_cases", **out)


# ---------------------------------------------------------------- G4 DIRECT known answers
def g4_direct():
    S = Shekel5(maximize=False)
    d = direct(S.f, S.bounds, maxiter=20)
    c = cdirect(S.f, S.bounds, maxiter=20)

    def foo(x, a1, a2):
        return -np.sum(np.sin(np.array(x) * a1) + np.array(x) * a2)
    b = [[0., 5.]] * 3
    c1 = cdirect(foo, b, args=[3.0, 0.0], maxiter=10)
    c2 = cdirect(foo, b, args=[-2.0, 2.0], maxiter=10)
    d1 = direct(foo, b, args=[3.0, 0.0], maxiter=20)
    # fixed-dimension behaviour: count samples with a wrapper
    cnt = [0]

    def foo3(x):
        cnt[0] += 1
        return float(np.sum((np.array(x) - .3) ** 2))
    counts = []
    for bb in ([[0., 1.]] * 3, [[0., 1.], [.5, .5], [0., 1.]], [[.5, .5], [0., 1.], [0., 1.]]):
        cnt[0] = 0
        r = cdirect(foo3, bb, maxiter=50, maxsample=10000)
        counts.append([cnt[0], r[0]] + list(r[1]))
    save("g4_direct", shekel_bounds=np.array(S.bounds, float), shekel_A=S.A[:5], shekel_C=S.C[:5],
         shekel_direct=np.r_[d[0], d[1]], shekel_cdirect=np.r_[c[0], c[1]],
         foo_c1=np.r_[c1[0], c1[1]], foo_c2=np.r_[c2[0], c2[1]], foo_d1=np.r_[d1[0], d1[1]],
         fixed_counts=np.array(counts))


# ---------------------------------------------------------------- G6 synthetic sweeps
def synth(seed, N, D):
    rs = np.random.RandomState(seed)
    X = rs.rand(N, D)
    Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    return X, Y


def g6_sweeps():
    out = {}
    names = []
    specs = [("c1_n32_d2_ard", 1, 32, 2, GaussianKernel_ard([.3, .3]), 1024),
             ("c2r_n256_d4_ard", 2, 256, 4, GaussianKernel_ard([.3] * 4), 512),
             ("c2r_n256_d4_iso", 2, 256, 4, GaussianKernel_iso([.3]), 256),
             ("c3r_n192_d8_m5", 3, 192, 8, MaternKernel5([.5, 1.0]), 256),
             ("c3r_n192_d8_m3", 3, 192, 8, MaternKernel3([.5, 1.0]), 256),
             ("c2_n1024_d4_ard", 2, 1024, 4, GaussianKernel_ard([.3] * 4), 64)]
    for name, seed, N, D, kernel, M in specs:
        X, Y = synth(seed, N, D)
        GP = GaussianProcess(kernel, X, Y, noise=.1)
        cand = np.random.RandomState(100 + seed).rand(M, D)
        post = np.array([GP.posterior(q) for q in cand])
        ei_py, pi_py, ucb_py = py_point_values(GP, cand, NA=D)
        p = name + "/"
        out.update({p + "seed": seed, p + "N": N, p + "D": D, p + "M": M, p + "ktype": ktype_of(kernel),
                    p + "hyper": np.array(kernel.hyperparams, float), p + "mu": post[:, 0], p + "s2": post[:, 1],
                    p + "ei_py": ei_py, p + "pi_py": pi_py, p + "ucb_py": ucb_py})
        if ktype_of(kernel) != 3:   # libego's Matern-5/2 branch is broken (stdout spam + OOB read)
            out[p + "ei_c"] = native_point_values(GP, cand, 0, .01)
            out[p + "pi_c"] = native_point_values(GP, cand, 1, .01)
            out[p + "ucb_c"] = native_point_values(GP, cand, 2, 1.5)
        names.append(name)
    out["names"] = np.array(names)
    save("g6_sweeps", **out)


# ---------------------------------------------------------------- G7 preference GPs
def g7_prefs():
    out = {}
    names = []
    tf = Hartman6()
    for P in (8, 16, 32, 64):
        rs = np.random.RandomState(40 + P)
        pts = rs.rand(2 * P, 6)
        prefs = []
        for i in range(P):
            a, b = pts[2 * i], pts[2 * i + 1]
            prefs.append((a, b, 0) if tf.f(a) > tf.f(b) else (b, a, 0))
        kernel = GaussianKernel_ard(tf.defaultHP[GaussianKernel_ard])
        GP = PrefGaussianProcess(kernel)
        GP.addPreferences(prefs)
        probe = np.random.RandomState(140 + P).rand(32, 6)
        post = np.array([GP.posterior(q) for q in probe])
        ei_c = native_point_values(GP, probe, 0, .01)
        # gallery with injected LHC samples
        inj = [np.random.RandomState(240 + P + r).rand(300, 6) for r in range(4)]
        it = iter(inj)
        refgallery.lhcSample = lambda bounds, n, seed=None: list(next(it))
        gal = refgallery.fastUCBGallery(GP, tf.bounds, 4)
        refgallery.lhcSample = lhcSample
        p = "p%d/" % P
        out.update({p + "pref_v": np.array([v for v, u, d in prefs]), p + "pref_u": np.array([u for v, u, d in prefs]),
                    p + "pref_d": np.array([d for v, u, d in prefs], float),
                    p + "hyper": np.array(kernel.hyperparams, float),
                    p + "X": GP.X, p + "Y": GP.Y, p + "C": GP.C, p + "R": GP.R, p + "L": GP.L,
                    p + "probe": probe, p + "post": post, p + "ei_c": ei_c,
                    p + "lhc": np.array(inj), p + "gallery": np.array(gal), p + "bounds": np.array(tf.bounds, float)})
        names.append("p%d" % P)
    out["names"] = np.array(names)
    save("g7_prefs", **out)


# ---------------------------------------------------------------- G8 NLML values
def g8_nlml():
    out = {}
    for N in (64, 256):
        X, Y = synth(5, N, 16)
        thetas = np.exp(np.random.RandomState(105).uniform(np.log(.5), np.log(3), size=(4, 16)))
        vals = []
        for th in thetas:
            vals.append(marginalLikelihood(GaussianKernel_ard(th), X, Y, 16, computeGradient=False, noise=1e-3))
        out["n%d_theta" % N] = thetas
        out["n%d_nlml" % N] = np.array(vals)
    X, Y = synth(5, 64, 16)
    v, g = marginalLikelihood(GaussianKernel_ard(np.full(16, 1.5)), X, Y, 16, computeGradient=True, noise=1e-3)
    out["n64_grad_theta"] = np.full(16, 1.5); out["n64_grad_nlml"] = v; out["n64_grad"] = g
    save("g8_nlml", **out)


# ---------------------------------------------------------------- G9 latin hypercube
def g9_lhc():
    out = {}
    for seed, bounds, n in ((22, [[0., 1.]], 5), (23, [[0., 5.], [0., 5.]], 5), (0, [[0., 10.]] * 4, 10),
                            (7, [[1., 1.], [-1.99, 1.98], [-1.99, 1.98]], 12)):
        out["s%d" % seed] = np.array(lhcSample(bounds, n, seed=seed))
        out["s%d_bounds" % seed] = np.array(bounds, float)
    save("g9_lhc", **out)


# ---------------------------------------------------------------- G10 the analytic test-function zoo
def g10_testfunctions():
    """f(x) of every analytic class of ego/utils/testfunctions.py on seeded points inside its bounds, both signs.
    Levy cannot be constructed in the reference (its __init__ reads an undefined name, testfunctions.py:314): its
    values are taken from the class's own f with the attribute the constructor meant to set."""
    import ego.utils.testfunctions as T
    out = {}
    names = []
    cases = [("Poly4", T.Poly4, {}), ("Poly6", T.Poly6, {}), ("Schubert1", T.Schubert1, {}),
             ("GoldsteinPrice", T.GoldsteinPrice, {}), ("Shekel5", T.Shekel5, {}), ("Shekel7", T.Shekel7, {}),
             ("Shekel10", T.Shekel10, {}), ("Camelback", T.Camelback, {}), ("Branin", T.Branin, {}),
             ("Hartman3", T.Hartman3, {}), ("Hartman6", T.Hartman6, {}),
             ("Michalewics2", T.Michalewics, dict(d=2)), ("Michalewics5", T.Michalewics, dict(d=5)),
             ("Michalewics10", T.Michalewics, dict(d=10)), ("Perm4", T.Perm, dict(d=4)), ("Perm3", T.Perm, dict(d=3)),
             ("Sphere4", T.Sphere, dict(d=4)), ("SumSquares4", T.SumSquares, dict(d=4)), ("SumSquares8", T.SumSquares, dict(d=8)),
             ("Zakharov2", T.Zakharov, dict(d=2)), ("Zakharov5", T.Zakharov, dict(d=5))]
    rs = np.random.RandomState(1010)
    for name, cls, kw in cases:
        tf = cls(maximize=False, **kw)
        b = np.array(tf.bounds, dtype=float)
        P = b[:, 0] + (b[:, 1] - b[:, 0]) * rs.rand(24, len(b))
        out[name + "_x"] = P
        out[name + "_f"] = np.array([float(tf.f(x)) for x in P])
        out[name + "_fmax"] = np.array([float(cls(maximize=True, **kw).f(x)) for x in P])
        out[name + "_bounds"] = b
        out[name + "_min"] = float(tf.minimum)
        out[name + "_name"] = np.array(tf.name)
        names.append(name)
    for d in (2, 4):                                # Levy: bypass the broken constructor
        tf = T.Levy.__new__(T.Levy)
        T.TestFunction.__init__(tf, "Levy %d" % d, 0, np.ones(d), [[-10.0, 10.0]] * d, maximize=False)
        tf.d = d
        b = np.array(tf.bounds, dtype=float)
        P = b[:, 0] + (b[:, 1] - b[:, 0]) * rs.rand(24, d)
        name = "Levy%d" % d
        out[name + "_x"] = P
        out[name + "_f"] = np.array([float(tf.f(x)) for x in P])
        tf.maximize = True
        out[name + "_fmax"] = np.array([float(tf.f(x)) for x in P])
        out[name + "_bounds"] = b
        out[name + "_min"] = 0.0
        out[name + "_name"] = np.array("Levy %d" % d)
        names.append(name)
    out["names"] = np.array(names)
    save("g10_testfunctions", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g6", "g7", "g8", "g9", "g10"]
    fns = dict(g1=g1_demo, g2=g2_hyper, g3=g3_cases, g4=g4_direct, g6=g6_sweeps, g7=g7_prefs, g8=g8_nlml, g9=g9_lhc,
               g10=g10_testfunctions)
    for w in which:
        fns[w]()
    shutil.rmtree(TMP, ignore_errors=True)
