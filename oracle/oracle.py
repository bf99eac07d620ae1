"""
oracle.py -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement of the reference's GP-posterior + acquisition hot path:

* the native half (cpp/libego: posterior, EI/PI/UCB, DIRECT) lives in
  oracle/ibo_oracle.c and is reached here through ctypes;
* the Python half (kernels, fit, posterior, acquisition classes, Python DIRECT,
  latin hypercube, marginal likelihood, preference GP, gallery) is restated
  below in NumPy, each function citing the reference lines it follows
  (paths relative to /root/reference).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this module.  ibo_amd/ never does: the product fails loudly without its HIP
library instead of falling back to anything in here.

Parity status: PINNED against golden vectors generated from the real reference
(tests/golden/make_golden.py) and, where oracle/_ref/libego.so exists, against
the reference's own compiled C++ (tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int, c_long, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DP = POINTER(c_double)

K_SE_ARD, K_SE_ISO, K_MATERN3, K_MATERN5 = 0, 1, 2, 3
ACQ_EI, ACQ_PI, ACQ_UCB = 0, 1, 2
ERF_LIBM, ERF_NR = 0, 1
CLAMP_NATIVE = 1e-8   # cpp/optimizeGP.cpp:150-153
CLAMP_PY = 10e-8      # ego/gaussianprocess/__init__.py:224


def _dp(a):
    return a.ctypes.data_as(_DP)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------
# library loading
# --------------------------------------------------------------------------
_lib = None


def build():
    """(Re)build oracle/_build/liboracle.so and, if /root/reference exists,
    oracle/_ref/libego.so.  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build()
        # (tools/sanitize_oracle.sh points this at a -fsanitize build of the same source)
        path = os.environ.get("IBO_ORACLE_LIB") or path
        L = ctypes.CDLL(path)
        L.orc_cov.restype = c_double
        L.orc_cov.argtypes = [c_int, c_int, _DP, _DP, _DP, c_double]
        L.orc_build_R.restype = None
        L.orc_build_R.argtypes = [c_int, c_int, c_int, _DP, _DP, c_double, c_double, _DP]
        L.orc_cov_matrix.restype = None
        L.orc_cov_matrix.argtypes = [c_int, c_int, c_int, _DP, _DP, c_double, _DP]
        L.orc_cholesky.restype = c_int
        L.orc_cholesky.argtypes = [c_int, _DP, _DP]
        L.orc_prior_mu.restype = c_double
        L.orc_prior_mu.argtypes = [c_int, _DP, c_int, _DP, _DP, c_double, _DP, _DP]
        L.orc_erf_nr.restype = c_double
        L.orc_erf_nr.argtypes = [c_double]
        L.orc_acq_value.restype = c_double
        L.orc_acq_value.argtypes = [c_int, c_int, c_double, c_double, c_double, c_double]
        L.orc_sweep_native.restype = c_int
        L.orc_sweep_native.argtypes = [c_int, _DP, _DP, _DP, c_int, c_int, c_int, _DP, c_double,
                                       c_int, _DP, _DP, c_double, _DP, _DP, c_double, c_double,
                                       c_int, c_double, c_long, _DP, _DP, _DP, _DP, _DP,
                                       POINTER(c_long)]
        L.orc_sweep_fast.restype = c_int
        L.orc_sweep_fast.argtypes = [c_int, _DP, _DP, _DP, c_int, c_int, c_int, _DP, c_double, c_double, c_double,
                                     c_int, c_double, c_double, c_long, _DP, _DP, _DP, POINTER(c_long), POINTER(c_int)]
        L.orc_posterior_chol.restype = c_int
        L.orc_posterior_chol.argtypes = [c_int, c_int, c_int, _DP, _DP, _DP, _DP, c_double, c_double,
                                         c_int, _DP, _DP, c_double, _DP, _DP, c_long, _DP, _DP, _DP]
        L.orc_direct.restype = c_void_p
        L.orc_acqmax_gp.restype = c_void_p
        L.orc_acqmax_gp.argtypes = [c_int, _DP, _DP, _DP, _DP, _DP, c_int, c_int, c_int, _DP, c_double,
                                    c_int, _DP, _DP, c_double, _DP, _DP, c_double, c_double,
                                    c_int, c_double, c_int, c_int, c_int, POINTER(c_long)]
        L.orc_nlml.restype = c_double
        L.orc_nlml.argtypes = [c_int, c_int, c_int, _DP, _DP, _DP, c_double, c_double]
        L.orc_free.restype = None
        L.orc_free.argtypes = [c_void_p]
        _lib = L
    return _lib


# --------------------------------------------------------------------------
# kernels  (ego/gaussianprocess/kernel.py)
# --------------------------------------------------------------------------
class Kern(object):
    """Kernel description.  kind in {'ard','iso','m3','m5','svard','sviso'}.

    ktype follows cdirectGP's isinstance chain (ego/acquisition/__init__.py:323-333):
    SV kernels are subclasses of their base and map to the base's code.
    sf2_py is what the Python model applies (kernel.py:66-68,203,243);
    sf2_native is what libego applies (cpp/optimizeGP.cpp:303-314): 1 for
    types 0-2, magnitude^2 for Matern-5/2 (the value the out-of-range
    hyperparams[ndim] read was meant to fetch)."""

    def __init__(self, kind, hyper):
        self.kind = kind
        self.hyper = np.array(hyper, dtype=float)
        if kind in ('ard', 'svard'):
            th = self.hyper if kind == 'ard' else self.hyper[:-1]
            self.theta = np.clip(th, 1e-4, 1e4)            # kernel.py:141
            self.ktype = K_SE_ARD
        elif kind in ('iso', 'sviso'):
            self.theta = self.hyper[:1].copy()
            self.ktype = K_SE_ISO
        elif kind == 'm3':
            self.theta = self.hyper[:1].copy()
            self.ktype = K_MATERN3
        elif kind == 'm5':
            self.theta = self.hyper[:1].copy()
            self.ktype = K_MATERN5
        else:
            raise ValueError(kind)
        if kind in ('svard', 'sviso', 'm3', 'm5'):
            self.sf2_py = float(np.exp(2.0 * np.log(self.hyper[-1])))
        else:
            self.sf2_py = 1.0
        self.sf2_native = self.sf2_py if kind == 'm5' else 1.0
        # what is handed to the native side as `hyperparams`
        self.c_hyper = _f64(self.hyper)

    # hyper vector in the layout orc_cov expects (length scales first)
    def oracle_hyper(self):
        if self.ktype == K_SE_ARD:
            return _f64(self.theta)
        return _f64(self.hyper)

    def cov(self, x1, x2):
        """scalar covariance, Python-model semantics (kernel.py cov methods)"""
        x1 = np.asarray(x1, dtype=float); x2 = np.asarray(x2, dtype=float)
        if self.ktype == K_SE_ARD:
            v = np.exp(-.5 * np.sum((1.0 / self.theta ** 2) * (x1 - x2) ** 2))
        elif self.ktype == K_SE_ISO:
            v = np.exp(-.5 * np.linalg.norm(x1 - x2) ** 2 * (1 / self.theta[0] ** 2))
        elif self.ktype == K_MATERN3:
            z = np.sqrt(3) * np.linalg.norm(x1 - x2) / self.theta[0]
            return self.sf2_py * (1.0 + z) * np.exp(-z)
        else:
            z = np.sum((np.sqrt(5.0) * (x1 - x2) / self.theta[0]) ** 2.0)
            return self.sf2_py * np.exp(-np.sqrt(z)) * (1.0 + np.sqrt(z) + z / 3.0)
        return self.sf2_py * v

    def cov_matrix(self, X):
        """kernel.py:46-53"""
        X = _f64(np.vstack(X))
        N, D = X.shape
        K = np.empty((N, N))
        lib().orc_cov_matrix(self.ktype, N, D, _dp(X), _dp(self.oracle_hyper()), self.sf2_py, _dp(K))
        return K

    def derivative(self, X, hp):
        """dK/dlog(theta_hp): kernel.py:92-106,122-127,152-166,183-188,212-227,251-266"""
        X = _f64(np.vstack(X))
        N, D = X.shape
        K = self.cov_matrix(X)
        diff = X[:, None, :] - X[None, :, :]
        nth = len(self.theta) if self.ktype == K_SE_ARD else 1
        if self.kind in ('svard', 'sviso') and hp == nth:
            return 2.0 * K
        if self.ktype == K_SE_ARD:
            if hp >= D:
                raise ValueError
            C = (1.0 / self.theta[hp] ** 2) * diff[:, :, hp] ** 2
            return K * C
        if self.ktype == K_SE_ISO:
            if hp != 0:
                raise ValueError
            C = np.sum(diff ** 2, axis=2) * (1 / self.theta[0] ** 2)
            return K * C
        if self.ktype == K_MATERN3:
            if hp == 0:
                r = np.sqrt(np.sum(diff ** 2, axis=2))
                C = self.sf2_py * r ** 2 * np.exp(-r)
                np.fill_diagonal(C, 0.0)
                return C
            if hp == 1:
                return 2.0 * K
            raise ValueError
        if hp == 0:
            z = np.sum((np.sqrt(5.0) * diff / self.theta[0]) ** 2.0, axis=2)
            C = self.sf2_py * (z + np.sqrt(z) ** 3.0) * np.exp(-np.sqrt(z)) / 3.0
            np.fill_diagonal(C, 0.0)
            return C
        if hp == 1:
            return 2.0 * K
        raise ValueError


# --------------------------------------------------------------------------
# RBFN mean prior (ego/gaussianprocess/prior.py:60-66)
# --------------------------------------------------------------------------
class Prior(object):
    def __init__(self, means, beta, theta, lowerb, width):
        self.means = _f64(np.atleast_2d(means))
        self.beta = _f64(beta)
        self.theta = float(theta)
        self.lowerb = _f64(lowerb)
        self.width = _f64(width)

    def mu(self, x):
        x = _f64(x).reshape(-1)
        return lib().orc_prior_mu(len(x), _dp(x), len(self.beta), _dp(self.means), _dp(self.beta),
                                  self.theta, _dp(self.lowerb), _dp(self.width))

    def cargs(self):
        return (len(self.beta), _dp(self.means), _dp(self.beta), self.theta, _dp(self.lowerb), _dp(self.width))


_DUMMY = np.zeros(1)


def _prior_cargs(prior):
    if prior is None:
        return (0, _dp(_DUMMY), _dp(_DUMMY), 0.0, _dp(_DUMMY), _dp(_DUMMY))
    return prior.cargs()


# --------------------------------------------------------------------------
# latin hypercube (ego/utils/latinhypercube.py:27-46)
# --------------------------------------------------------------------------
def lhc_sample(bounds, N, seed=None):
    rs = np.random.RandomState(seed)
    samp = []
    for bmin, bmax in bounds:
        if bmin == bmax:
            dsamp = np.array([bmin] * N)
        else:
            dsamp = (bmax - bmin) * rs.rand(N) / N + np.arange(bmin, bmax, (bmax - bmin) / N)
        rs.shuffle(dsamp)
        samp.append(dsamp)
    return list(np.vstack(samp).T)


# --------------------------------------------------------------------------
# GP model, Python-path semantics (ego/gaussianprocess/__init__.py:81-328)
# --------------------------------------------------------------------------
class GP(object):
    def __init__(self, kern, X, Y, noise=.1, prior=None):
        self.kern, self.noise, self.prior = kern, float(noise), prior
        self.X = _f64(np.atleast_2d(X))
        self.Y = _f64(np.atleast_1d(Y)).reshape(-1)
        N, D = self.X.shape
        self.R = np.empty((N, N))
        # _computeCorrelations :134-149 (diagonal 1+noise; kernel never called for i==j)
        lib().orc_build_R(kern.ktype, N, D, _dp(self.X), _dp(kern.oracle_hyper()), kern.sf2_py,
                          self.noise, _dp(self.R))
        self.L = np.linalg.cholesky(self.R)          # :299
        self.M = None                                 # PrefGP: matrix that replaces R

    def factor_matrix(self):
        return self.R if self.M is None else self.M

    def posteriors(self, Q):
        """(mu, sigma2) arrays, Python-path clamp [1e-7, 10] (:169-244)"""
        Q = _f64(np.atleast_2d(Q))
        M = len(Q)
        mu = np.empty(M); s2 = np.empty(M)
        N, D = self.X.shape
        L = _f64(self.L)
        lib().orc_posterior_chol(self.kern.ktype, D, N, _dp(self.X), _dp(self.Y), _dp(L),
                                 _dp(self.kern.oracle_hyper()), self.kern.sf2_py, self.noise,
                                 *_prior_cargs(self.prior), M, _dp(Q), _dp(mu), _dp(s2))
        return mu, s2

    def posterior(self, x):
        m, v = self.posteriors(np.asarray(x, dtype=float).reshape(1, -1))
        return m[0], v[0]

    def mu(self, x):
        return self.posterior(x)[0]

    # ---- native path (what cdirectGP hands to libego) ----
    def inv_factor(self):
        """ego/acquisition/__init__.py:385-388"""
        return _f64(np.linalg.inv(self.factor_matrix()))


def ucb_parm_native(nY, NA, delta, scale):
    """ego/acquisition/__init__.py:316-319 (NA/2 is Python-2 integer division)"""
    t = nY + 1
    return float(np.sqrt(scale * 2.0 * np.log(t ** (NA // 2 + 2) * np.pi ** 2 / (3.0 * delta))))


def ucb_coef_py(nY, NA, delta, scale):
    """UCB class: sqrt(scale*sBeta), sBeta=sqrt(2 log ...) (ego/acquisition/__init__.py:61-71)"""
    t = nY + 1
    sBeta = np.sqrt(2.0 * np.log(t ** (NA // 2 + 2) * np.pi ** 2 / (3.0 * delta)))
    return float(np.sqrt(scale * sBeta))


def acq_value(acq, erf_mode, mu, sigma, maxY, parm):
    f = lib().orc_acq_value
    mu = np.atleast_1d(mu); sigma = np.atleast_1d(sigma)
    return np.array([f(acq, erf_mode, float(m), float(s), float(maxY), float(parm))
                     for m, s in zip(mu, sigma)])


def sweep_native(gp, cand, acq=ACQ_EI, parm=0.01, erf_mode=ERF_LIBM, clamp_lo=CLAMP_NATIVE,
                 sf2=None, invR=None):
    """M independent native posterior+acquisition calls (cpp/optimizeGP.cpp:57-236).
    Returns dict(mu, s2, acq, best_val, best_idx)."""
    cand = _f64(np.atleast_2d(cand))
    M, D = cand.shape
    N = len(gp.Y)
    invR = gp.inv_factor() if invR is None else _f64(invR)
    mu = np.empty(M); s2 = np.empty(M); av = np.empty(M)
    bv = c_double(); bi = c_long()
    sf2 = gp.kern.sf2_native if sf2 is None else sf2
    rc = lib().orc_sweep_native(D, _dp(invR), _dp(gp.X), _dp(gp.Y), N, acq, gp.kern.ktype,
                                _dp(gp.kern.oracle_hyper()), sf2, *_prior_cargs(gp.prior),
                                float(parm), gp.noise, erf_mode, clamp_lo, M, _dp(cand),
                                _dp(mu), _dp(s2), _dp(av), ctypes.byref(bv), ctypes.byref(bi))
    assert rc == 0
    return dict(mu=mu, s2=s2, acq=av, best_val=bv.value, best_idx=bi.value)


def effective_cores():
    """CPUs this process may really use: the affinity mask cut down to the cgroup's CPU quota (a container often sees every CPU
    of the host but is throttled to a few cores' worth of time; OpenMP would start one thread per visible CPU)"""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(float(txt[0]) / float(txt[1])))))
            else:
                q = float(txt[0])
                if q > 0:
                    per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(math.ceil(q / per))))
            break
        except Exception:
            continue
    return n


def sweep_fast(gp, cand, acq=ACQ_EI, parm=0.01, erf_mode=ERF_LIBM, clamp_lo=CLAMP_NATIVE, threads=None):
    """best-effort all-core CPU sweep (alpha cached, triangular W = L^-1, OpenMP): same values as
    sweep_native up to rounding.  Returns dict(acq, best_val, best_idx, threads)."""
    assert gp.prior is None
    cand = _f64(np.atleast_2d(cand))
    M, D = cand.shape
    N = len(gp.Y)
    W = _f64(np.linalg.inv(np.linalg.cholesky(gp.factor_matrix())))
    alpha = _f64(W.T.dot(W.dot(gp.Y)))
    av = np.empty(M); bv = c_double(); bi = c_long(); nt = c_int(effective_cores() if threads is None else int(threads))
    rc = lib().orc_sweep_fast(D, _dp(W), _dp(alpha), _dp(gp.X), N, acq, gp.kern.ktype, _dp(gp.kern.oracle_hyper()),
                              gp.kern.sf2_native, float(parm), gp.noise, erf_mode, clamp_lo, float(np.max(gp.Y)), M,
                              _dp(cand), _dp(av), ctypes.byref(bv), ctypes.byref(bi), ctypes.byref(nt))
    assert rc == 0
    return dict(acq=av, best_val=bv.value, best_idx=bi.value, threads=nt.value)


def acqmax_native(gp, bounds, acq=ACQ_EI, parm=0.01, maxiter=50, maxtime=30, maxsample=10000,
                  erf_mode=ERF_LIBM, clamp_lo=CLAMP_NATIVE, invR=None):
    """cdirectGP + acqmaxGP (ego/acquisition/__init__.py:307-447, cpp/optimizeGP.cpp:262-349).
    Returns (opt, optx, nsamples)."""
    lb = _f64([b[0] for b in bounds]); ub = _f64([b[1] for b in bounds])
    D = len(lb)
    invR = gp.inv_factor() if invR is None else _f64(invR)
    ns = c_long()
    p = lib().orc_acqmax_gp(D, _dp(lb), _dp(ub), _dp(invR), _dp(gp.X), _dp(gp.Y), len(gp.Y), acq,
                            gp.kern.ktype, _dp(gp.kern.oracle_hyper()), gp.kern.sf2_native,
                            *_prior_cargs(gp.prior), float(parm), gp.noise, erf_mode, clamp_lo,
                            maxiter, maxtime, maxsample, ctypes.byref(ns))
    res = np.ctypeslib.as_array(ctypes.cast(p, _DP), shape=(D + 1,)).copy()
    lib().orc_free(p)
    return -res[0], res[1:], ns.value


_OBJ = ctypes.CFUNCTYPE(c_double, c_int, _DP)


def cdirect(f, bounds, args=None, maxiter=10, maxtime=10, maxsample=200000):
    """native DIRECT on a Python objective (ego/utils/optimize.py:310-343).
    Returns (fmin, xmin, nsamples)."""
    args = [] if args is None else args
    lb = _f64([b[0] for b in bounds]); ub = _f64([b[1] for b in bounds])
    n = len(lb)

    def obj(k, x):
        return float(f(np.array([x[i] for i in range(k)]), *args))
    cb = _OBJ(obj)
    ns = c_long()
    L = lib()
    L.orc_direct.argtypes = [_OBJ, c_int, _DP, _DP, c_int, c_int, c_int, POINTER(c_long)]
    p = L.orc_direct(cb, n, _dp(lb), _dp(ub), maxiter, maxtime, maxsample, ctypes.byref(ns))
    res = np.ctypeslib.as_array(ctypes.cast(p, _DP), shape=(n + 1,)).copy()
    L.orc_free(p)
    return res[0], res[1:], ns.value


# --------------------------------------------------------------------------
# Python DIRECT (ego/utils/optimize.py:58-280).  The reference keeps its
# rectangles in a set (iteration order = object hashes); a list is used here,
# which changes nothing but tie order.
# --------------------------------------------------------------------------
class _PyRect(object):
    def __init__(self, lb, ub, y):
        self.lb = list(lb); self.ub = list(ub); self.y = y
        # optimize.py:63 zips (lb, ub) into (u, l): centre = ub + (lb-ub)/2
        self.center = [l + (u - l) / 2. for u, l in zip(self.lb, self.ub)]
        self.d = sum([(l - c) ** 2. for l, c in zip(self.lb, self.center)]) ** 0.5


def direct_py(f, bounds, args=None, maxiter=None, maxsample=None, maxtime=None):
    """Returns (fmin, xmin, nsamples)."""
    import time as _t
    if not (maxiter or maxsample or maxtime):
        raise ValueError("No termination criterion set!")
    args = [] if args is None else args
    st = dict(fmin=None, samples=0)
    tic = _t.time()

    def samplef(x):
        xp = [z * (b[1] - b[0]) + b[0] for z, b in zip(x, bounds)]
        y = f(xp, *args)
        st['samples'] += 1
        if st['fmin'] is None or y < st['fmin'][0]:
            st['fmin'] = [y, list(x)]
        return y

    N = len(bounds)
    rects = []

    def divrec(rect):
        rects.remove(rect)
        maxlength = max([u - l for u, l in zip(rect.ub, rect.lb)])
        I = []
        for i in range(N):
            if rect.ub[i] - rect.lb[i] == maxlength:
                s1 = list(rect.center); s2 = list(rect.center)
                w = rect.ub[i] - rect.lb[i]
                s1[i] = rect.lb[i] + w / 3.
                s2[i] = rect.lb[i] + 2. * w / 3.
                I.append((i, min(samplef(s1), samplef(s2))))
        I.sort(key=lambda t: t[1])
        old = rect
        target = rect
        for i, _ in I:
            w = old.ub[i] - old.lb[i]
            split1 = old.lb[i] + w * (1 / 3)
            split2 = old.lb[i] + w * (2 / 3)
            lb1 = list(old.lb); ub1 = list(old.ub); ub1[i] = split1
            rects.append(_PyRect(lb1, ub1, samplef([l + (u - l) / 2. for u, l in zip(lb1, ub1)])))
            lb2 = list(old.lb); ub2 = list(old.ub); lb2[i] = split1; ub2[i] = split2
            target = _PyRect(lb2, ub2, old.y)
            lb3 = list(old.lb); ub3 = list(old.ub); lb3[i] = split2
            rects.append(_PyRect(lb3, ub3, samplef([l + (u - l) / 2. for u, l in zip(lb3, ub3)])))
            old = target
        rects.append(target)

    def results():
        lo = np.array([b[0] for b in bounds], dtype=float)
        hi = np.array([b[1] for b in bounds], dtype=float)
        return st['fmin'][0], np.array(st['fmin'][1]) * (hi - lo) + lo, st['samples']

    first = _PyRect([0.] * N, [1.] * N, samplef([.5] * N))
    rects.append(first)
    divrec(first)
    it = 0
    eps = 10e-10
    while True:
        it += 1
        if maxiter and it > maxiter:
            return results()
        pot = []
        for Rj in list(rects):
            maxI1 = None; minI2 = None; broke = False
            for Ri in rects:
                if Ri is Rj:
                    continue
                if Ri.d < Rj.d:
                    v = (Rj.y - Ri.y) / (Rj.d - Ri.d)
                    if maxI1 is None or v > maxI1:
                        maxI1 = v
                elif Ri.d > Rj.d:
                    v = (Ri.y - Rj.y) / (Ri.d - Rj.d)
                    if minI2 is None or v < minI2:
                        minI2 = v
                        if minI2 <= 0.:
                            broke = True; break
                else:
                    if Rj.y > Ri.y:
                        broke = True; break
                if maxI1 is not None and minI2 is not None and minI2 < maxI1:
                    broke = True; break
            if not broke:
                F = st['fmin'][0]
                if not minI2:
                    pot.append(Rj)
                elif F == 0:
                    if Rj.y <= Rj.d * minI2:
                        pot.append(Rj)
                elif eps <= (F - Rj.y) / abs(F) + (Rj.d / abs(F)) * minI2:
                    pot.append(Rj)
            if maxtime and _t.time() - tic >= maxtime:
                return results()
        for Rj in pot:
            divrec(Rj)
            if maxsample and st['samples'] >= maxsample:
                return results()
            if maxtime and _t.time() - tic >= maxtime:
                return results()


# --------------------------------------------------------------------------
# marginal likelihood (ego/gaussianprocess/trainhyper.py:47-95)
# --------------------------------------------------------------------------
def marginal_likelihood(kern, X, Y, nhyper, compute_gradient=True, noise=1e-3):
    X = _f64(np.vstack(X)); Y = _f64(Y)
    N = len(X)
    K = kern.cov_matrix(X) + np.eye(N) * noise
    L = np.linalg.cholesky(K)
    alpha = np.linalg.solve(L.T, np.linalg.solve(L, Y))
    nlml = 0.5 * np.dot(Y, alpha) + np.sum(np.log(np.diag(L))) + 0.5 * N * np.log(2.0 * np.pi)
    if not compute_gradient:
        return nlml
    W = np.linalg.solve(L.T, np.linalg.solve(L, np.eye(N))) - np.outer(alpha, alpha)
    d = np.array([np.sum(W * kern.derivative(X, i)) / 2.0 for i in range(nhyper)])
    return nlml, d


def nlml_c(kern, X, Y, noise=1e-3):
    """plain-C NLML (no gradient); NaN when K is not positive definite"""
    X = _f64(np.vstack(X)); Y = _f64(Y)
    N, D = X.shape
    return lib().orc_nlml(kern.ktype, N, D, _dp(X), _dp(Y), _dp(kern.oracle_hyper()), kern.sf2_py, noise)


# --------------------------------------------------------------------------
# preference GP (ego/gaussianprocess/__init__.py:331-498)
# --------------------------------------------------------------------------
def _cdf_nr(x):
    return 0.5 * (1 + lib().orc_erf_nr(float(x) * 0.707106))


def _pdf_nr(x):
    return float(np.exp(-(x ** 2 / 2)) * 0.398942)


def pref_index(prefs):
    """dedupe points in first-seen order (:389-408) -> (X, [(v,u,d)], set of preferred idx)"""
    x2ind = {}; inds = []; vs = set()
    for v, u, d in prefs:
        v = tuple(np.asarray(v, dtype=float)); u = tuple(np.asarray(u, dtype=float))
        vs.add(v)
        if v not in x2ind:
            x2ind[v] = len(x2ind)
        if u not in x2ind:
            x2ind[u] = len(x2ind)
        inds.append((x2ind[v], x2ind[u], d))
    X = np.array([x for x, _ in sorted(x2ind.items(), key=lambda t: t[1])], dtype=float)
    return X, inds, set(x2ind[v] for v in vs)


def pref_S(y, inds, L):
    """MAP functional (:351-385, useC=False branch -- the one fmin_bfgs calls)"""
    s = 0.
    Z = np.sqrt(2)
    for v, u, d in inds:
        s += (d + 1) * np.log(_cdf_nr((y[v] - y[u]) / Z) + 1e-10)
    Lx = np.linalg.solve(L, y)
    return -s + np.dot(Lx, Lx) / 2


def pref_C(gp, inds):
    """Laplace 'C' matrix (:459-486).  gp must already hold X, Y_map and L=chol(R)."""
    n = len(gp.X)
    C = np.eye(n) * 5
    mu = gp.posteriors(gp.X)[0]
    for v, u, _ in inds:
        d = (mu[v] - mu[u]) / (np.sqrt(2) * np.sqrt(gp.noise))
        cdf = max(_cdf_nr(d), 1e-10); pdf = max(_pdf_nr(d), 1e-10)
        w = 1.0 / (2 * gp.noise) * (pdf ** 2 / cdf ** 2 + d * pdf / cdf)
        C[v, u] -= w; C[u, v] -= w
        C[v, v] += w; C[u, u] += w
    return C


def pref_fit(kern, prefs, noise=.1, prior=None, Y_map=None):
    """addPreferences on an empty PrefGP (:347-498).  If Y_map is given the
    BFGS step is skipped (parity is pinned downstream of the MAP)."""
    from scipy.optimize import fmin_bfgs
    X, inds, vset = pref_index(prefs)
    gp = GP(kern, X, np.zeros(len(X)), noise=noise, prior=prior)
    if Y_map is None:
        start = [.5 if i in vset else -.5 for i in range(len(X))]
        Y = fmin_bfgs(pref_S, start, args=(inds, gp.L), disp=0)
        # order fix-up (:445-457)
        for v, u, _ in inds:
            if Y[v] <= Y[u]:
                if not any(u1 == v for _, u1, _ in inds):
                    Y[v] = Y[u] + .1
    else:
        Y = _f64(Y_map).copy()
    gp.Y = _f64(Y)
    gp.C = pref_C(gp, inds)
    Mx = gp.R + np.linalg.inv(gp.C)
    for _ in range(11):
        try:
            gp.L = np.linalg.cholesky(Mx)
            break
        except np.linalg.LinAlgError:
            gp.C += np.eye(len(X))
            Mx = gp.R + np.linalg.inv(gp.C)
    gp.M = Mx
    gp.inds = inds
    return gp


# --------------------------------------------------------------------------
# gallery (ego/acquisition/gallery.py:42-136) with the LHC samples injected
# --------------------------------------------------------------------------
def fast_gallery(gp, bounds, N, lhc_per_round, use_best=True, maxiter=50, maxsample=10000, fast=False):
    """gp: GP or pref GP with data.  lhc_per_round: list of (S,D) arrays, one per round.
    Returns (gallery list, trace list of dict(opt, optx, best_lhc)).
    fast=True evaluates the sample step with sweep_fast (all cores, alpha cached, triangular L^-1: the same
    posterior + NR-erf EI, pinned against the per-point path by tests/test_oracle_golden.py) -- what makes
    20 000-candidate rounds on 2000-point models affordable for the GPU parity tests; no prior mean."""
    gallery = []
    if use_best:
        bestY = -np.inf; bestX = None
        for x, y in zip(gp.X, gp.Y):
            if y > bestY and all(b[0] <= v <= b[1] for v, b in zip(x, bounds)):
                bestY, bestX = y, x
        if bestX is not None:
            gallery.append(np.array(bestX))
    h = GP(gp.kern, gp.X.copy(), gp.Y.copy(), prior=gp.prior)      # default noise .1 (:67)
    trace = []
    rnd = 0
    while len(gallery) < N:
        bestU = -np.inf; bestX = None
        opt, optx, _ = acqmax_native(h, bounds, ACQ_EI, parm=.3, maxiter=maxiter, maxsample=maxsample)
        if len(gallery) == 0 or min(np.linalg.norm(optx - g) for g in gallery) > .5:
            bestU, bestX = opt, optx
        S = _f64(lhc_per_round[rnd])
        if fast:
            # (the scan below in vector form: first index of the largest admissible value, and only if it beats DIRECT's)
            assert h.kern.sf2_native == h.kern.sf2_py
            u = sweep_fast(h, S, ACQ_EI, .4, ERF_NR, CLAMP_PY)["acq"]
            if len(gallery):
                G = np.array(gallery)
                dmin = np.min(np.sqrt(np.sum((S[:, None, :] - G[None, :, :]) ** 2, axis=2)), axis=1)
                u = np.where(dmin > .5, u, -np.inf)
            k = int(np.argmax(u))
            if u[k] > bestU:
                bestU, bestX = u[k], S[k]
        else:
            mu, s2 = h.posteriors(S)
            u = acq_value(ACQ_EI, ERF_NR, mu, np.sqrt(s2), np.max(h.Y), .4)
            for x, ux in zip(S, u):
                if ux > bestU and min(np.linalg.norm(x - g) for g in gallery) > .5:
                    bestU, bestX = ux, x
        gallery.append(np.array(bestX))
        trace.append(dict(opt=opt, optx=np.array(optx), chosen=np.array(bestX), u=bestU))
        h = GP(h.kern, np.vstack([h.X, bestX]), np.r_[h.Y, h.mu(bestX)], prior=h.prior)
        rnd += 1
    return gallery, trace


# --------------------------------------------------------------------------
# the reference's own compiled C++ (oracle/_ref/libego.so), when present
# --------------------------------------------------------------------------
class RefLib(object):
    """ctypes binding with exactly the argtypes cdirectGP declares
    (ego/acquisition/__init__.py:343-364)."""

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "_ref", "libego.so")
        self.lib = ctypes.CDLL(path)
        self.lib.acqmaxGP.restype = _DP
        self.lib.acqmaxGP.argtypes = [c_int, _DP, _DP, _DP, _DP, _DP, c_int, c_int, c_int, _DP, c_int,
                                      _DP, _DP, c_double, _DP, _DP, c_double, c_double,
                                      c_int, c_int, c_int]
        self.lib.direct.restype = _DP
        self.lib.direct.argtypes = [_OBJ, c_int, _DP, _DP, c_int, c_int, c_int]
        self.libc = ctypes.CDLL(None)
        self.libc.free.argtypes = [c_void_p]
        self.libc.free.restype = None

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libego.so"))

    def acqmax(self, gp, bounds, acq=ACQ_EI, parm=0.01, maxiter=50, maxtime=30, maxsample=10000, invR=None):
        lb = _f64([b[0] for b in bounds]); ub = _f64([b[1] for b in bounds])
        D = len(lb)
        invR = gp.inv_factor() if invR is None else _f64(invR)
        r = self.lib.acqmaxGP(D, _dp(lb), _dp(ub), _dp(invR), _dp(gp.X), _dp(gp.Y), len(gp.Y), acq,
                              gp.kern.ktype, _dp(gp.kern.c_hyper), *_prior_cargs(gp.prior),
                              float(parm), gp.noise, maxiter, maxtime, maxsample)
        res = np.array([r[i] for i in range(D + 1)])
        self.libc.free(r)
        return -res[0], res[1:]

    def direct(self, f, bounds, args=None, maxiter=10, maxtime=10, maxsample=200000):
        args = [] if args is None else args
        lb = _f64([b[0] for b in bounds]); ub = _f64([b[1] for b in bounds])
        n = len(lb)
        cnt = [0]

        def obj(k, x):
            cnt[0] += 1
            return float(f(np.array([x[i] for i in range(k)]), *args))
        r = self.lib.direct(_OBJ(obj), n, _dp(lb), _dp(ub), maxiter, maxtime, maxsample)
        res = np.array([r[i] for i in range(n + 1)])
        self.libc.free(r)
        return res[0], res[1:], cnt[0]
