"""
Gaussian-process models with the reference's API (ego/gaussianprocess/__init__.py)
on the MI355X backend.

    GaussianProcess(kernel, X=None, Y=None, prior=None, noise=.1, gnoise=1e-4, G=None)
        .addData(X, Y)  .posterior(x, getvar=True)  .posteriors(X)  .mu(x)  .negmu(x)
        .getYfromX(qx)  .done(x)   attributes X, Y, R, L, noise, kernel, prior, ...
    PrefGaussianProcess(kernel, prefs=None, **kw)
        .addPreferences(prefs)  .addObservationPoint(x)   attributes preferences, C

Everything numerical -- K(X,X), Cholesky, L^-1, posterior mean/variance -- runs
in libibo_hip on the GPU; this file holds the bookkeeping the reference keeps
in Python (data accumulation, preference indexing, the MAP optimiser's control
flow).  There is no NumPy fallback: without the library or a GPU, fitting raises.
"""
import ctypes
from time import time

import numpy as np

from .. import _lib
from .._lib import NotPositiveDefinite
from .kernel import GaussianKernel_ard, GaussianKernel_iso, MaternKernel3, MaternKernel5  # noqa: F401

LinAlgError = np.linalg.LinAlgError


# ---------------------------------------------------------------------------
# scalar helpers kept for API compatibility (ego/gaussianprocess/__init__.py:55-77):
# Numerical-Recipes erf (fractional error 1.2e-7) and the CDF/PDF built on it.
# The device twin is erf_nr_dev in csrc/ibo_common.h.
# ---------------------------------------------------------------------------
def erf(z):
    """works on scalars and on arrays (the preference GP evaluates it for every pair at once)"""
    t = 1.0 / (1.0 + 0.5 * np.abs(z))
    poly = 0.17087277
    for c in (-0.82215223, 1.48851587, -1.13520398, 0.27886807, -0.18628806, 0.09678418, 0.37409196, 1.00002368):
        poly = c + t * poly
    ans = 1 - t * np.exp(-z * z - 1.26551223 + t * poly)
    return np.where(np.asarray(z) >= 0.0, ans, -ans) if np.ndim(z) else (ans if z >= 0.0 else -ans)


def CDF(x):
    return 0.5 * (1 + erf(x * 0.707106))


def PDF(x):
    return np.exp(-(x ** 2 / 2)) * 0.398942


class _DeviceGP(object):
    """owner of one ibo_gp_t handle"""

    def __init__(self, device=None):
        self.device = _lib.default_device() if device is None else device
        h = ctypes.c_void_p()
        _lib.check(_lib.lib.ibo_gp_create(self.device, ctypes.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            try:
                _lib.lib.ibo_gp_destroy(self.h)
            except Exception:
                pass
            self.h = None

    def __del__(self):
        self.close()


class GaussianProcess(object):

    def __init__(self, kernel, X=None, Y=None, prior=None, noise=.1, gnoise=1e-4, G=None, device=None, reserve_rows=0):
        self.kernel = kernel
        self._reserve_rows = int(reserve_rows)      # head-room for addData's in-place extension (ibo_gp_reserve)
        self.prior = prior
        self.noise = noise
        self.gnoise = np.array(gnoise, ndmin=1)
        self._device = device
        self._dev = None            # _DeviceGP, created at the first fit
        self._cache = {}
        self._prior_key, self._prior_pushed = None, False

        if (X is None and Y is not None) or (X is not None and Y is None):
            raise ValueError

        self.X = np.zeros((0, 0))
        self.Y = np.zeros((0))
        self.G = None
        self.name = 'GP'
        self.starttime = time()

        if G is not None:
            raise NotImplementedError("gradient observations are dead code in the reference "
                                      "(covWithGradients is never defined) and are not supported")
        if X is not None:
            self.addData(X, Y)

        self.augR = None
        self.augL = None
        self.augX = None
        self._augdev = None
        self.selected = None
        self.endtime = None

    # ------------------------------------------------------------------ device plumbing
    def _handle(self):
        if self._dev is None:
            self._dev = _DeviceGP(self._device)
            if self._reserve_rows:
                _lib.check(_lib.lib.ibo_gp_reserve(self._dev.h, self._reserve_rows))
        return self._dev.h

    def _fit_device(self, A=None, dev=None, X=None):
        """(re)fit on the GPU: R, L = chol(R or A), W = L^-1, alpha (csrc/abi.hip fit_impl)"""
        X = self.X if X is None else X
        ktype, hyper, sf2, _ = self.kernel._ibo_spec()
        h = self._handle() if dev is None else dev.h
        N, D = X.shape
        Xc = _lib.f64(X)
        Yc = _lib.f64(self.Y) if len(self.Y) == N else np.zeros(N)
        info = ctypes.c_int(0)
        if A is None:
            rc = _lib.lib.ibo_gp_fit(h, ktype, N, D, _lib.dp(Xc), _lib.dp(Yc), _lib.dp(hyper), len(hyper), sf2,
                                     float(self.noise), ctypes.byref(info))
        else:
            Ac = _lib.f64(A)
            rc = _lib.lib.ibo_gp_fit_with_matrix(h, ktype, N, D, _lib.dp(Xc), _lib.dp(Yc), _lib.dp(hyper), len(hyper),
                                                 sf2, float(self.noise), _lib.dp(Ac), ctypes.byref(info))
        _lib.check(rc)
        if dev is None:
            self._cache = {}
            self._prior_pushed = False
            self._push_prior()
            # what the device factor was built from: only an unchanged plain model may be extended in place
            self._fit_spec = (ktype, tuple(hyper), sf2, float(self.noise)) if A is None and X is self.X else None

    def _prior_arrays(self):
        """flatten the mean prior for the device: (means (k, D), beta, theta, lowerb, width) or None.
        The device evaluates RBF-network priors (the only kind the reference's native path knows,
        cpp/optimizeGP.cpp:116-133); anything else is refused loudly rather than silently dropped."""
        p = self.prior
        if p is None:
            return None
        need = ("means", "beta", "theta", "lowerb", "width")
        if not all(hasattr(p, a) for a in need):
            raise NotImplementedError("the device path evaluates RBFNMeanPrior-style priors (means, beta, theta, lowerb, "
                                      "width); %s is not one" % type(p).__name__)
        if any(getattr(p, a) is None for a in need):
            raise ValueError("the mean prior has not been trained or filled in (means/beta/lowerb/width missing)")
        beta = _lib.f64(np.asarray(p.beta, dtype=float).reshape(-1))
        means = _lib.f64(np.array(p.means, dtype=float).reshape(len(beta), -1))
        return means, beta, float(p.theta), _lib.f64(p.lowerb), _lib.f64(p.width)

    def _push_prior(self):
        """bring the device's copy of the prior in line with self.prior.  Called before every device
        evaluation: the prior may have been assigned, trained or edited in place since the fit, and the
        arrays are a few hundred bytes (the comparison costs microseconds)."""
        if self._dev is None:
            return
        arrs = self._prior_arrays()
        key = None if arrs is None else (arrs[0].tobytes(), arrs[1].tobytes(), arrs[2], arrs[3].tobytes(), arrs[4].tobytes())
        if key == self._prior_key and self._prior_pushed:
            return
        h = self._handle()
        if arrs is None:
            _lib.check(_lib.lib.ibo_gp_set_prior(h, 0, None, None, 0.0, None, None))
        else:
            means, beta, theta, lo, wd = arrs
            if means.shape[1] != self.X.shape[1]:
                raise ValueError("prior centres have %d dimensions, the data %d" % (means.shape[1], self.X.shape[1]))
            _lib.check(_lib.lib.ibo_gp_set_prior(h, len(beta), _lib.dp(means), _lib.dp(beta), theta, _lib.dp(lo), _lib.dp(wd)))
        self._prior_key, self._prior_pushed = key, True

    def _get_matrix(self, name):
        if name not in self._cache:
            N = len(self.X)
            out = np.empty((N, N))
            fn = getattr(_lib.lib, "ibo_gp_get_" + name)
            _lib.check(fn(self._handle(), _lib.dp(out)))
            self._cache[name] = out
        return self._cache[name]

    @property
    def R(self):
        if len(self.X) == 0:
            return None
        return self._get_matrix("R")

    @R.setter
    def R(self, v):          # the reference initialises self.R = None
        if v is not None:
            self._cache["R"] = v

    @property
    def L(self):
        return self._get_matrix("L")

    @L.setter
    def L(self, v):
        self._cache["L"] = v

    def last_fit_ms(self):
        ms = ctypes.c_float(0)
        _lib.check(_lib.lib.ibo_gp_last_fit_ms(self._handle(), ctypes.byref(ms)))
        return ms.value

    def _eval(self, Q, acq=_lib.ACQ_NONE, parm=0.0, erf_mode=_lib.ERF_NR, clamp_lo=_lib.CLAMP_PY,
              want=("mu", "s2"), ymax=None):
        """host points in, host arrays out, through the device sweep (PCIe-inclusive)"""
        Q = _lib.f64(np.atleast_2d(Q))
        M, D = Q.shape
        self._push_prior()
        if self._augdev is not None and acq != _lib.ACQ_NONE:
            # augmented variance in force (addObservationPoint): mean and variance come from two
            # factors, so the acquisition is formed from them as the reference's classes do
            # (ego/acquisition/__init__.py:68-71,107-110,150-164), per point on the host
            mu, s2 = self._posterior_arrays(Q, True)
            ym = np.max(self.Y) if ymax is None else ymax
            sig = np.sqrt(s2)
            if acq == _lib.ACQ_UCB:
                val = mu + parm * sig
            else:
                yd = mu - ym - parm
                Z = yd / sig
                if acq == _lib.ACQ_PI:
                    val = CDF(Z)
                else:
                    val = yd * CDF(Z) + sig * PDF(Z)
            return {"mu": mu, "s2": s2, "acq": val, "best_val": float(np.max(val)), "best_idx": int(np.argmax(val))}
        res = {k: np.empty(M) for k in want}
        ptr = lambda k: _lib.dp(res[k]) if k in res else None
        _lib.check(_lib.lib.ibo_acq_batch(self._handle(), M, _lib.dp(Q), acq, float(parm), erf_mode, clamp_lo,
                                          float('nan') if ymax is None else float(ymax), ptr("mu"), ptr("s2"), ptr("acq")))
        if "acq" in res:                             # first maximiser, NaNs never win: what the device arg-max does
            score = np.where(np.isnan(res["acq"]), -np.inf, res["acq"])
            k = int(np.argmax(score))
            res["best_val"], res["best_idx"] = float(score[k]), k
        return res

    # ------------------------------------------------------------------ reference API
    def posterior(self, X, getvar=True):
        """posterior mean (and variance) at ONE point (ego/gaussianprocess/__init__.py:169-228)"""
        if len(self.X) == 0:
            if self.prior is None:
                return (0.0, 1.0) if getvar else 0.0
            return (self.prior.mu(X), 1.0) if getvar else self.prior.mu(X)
        X = np.array(X, dtype=float, ndmin=2)
        mu, s2 = self._posterior_arrays(X[:1], getvar)
        if getvar:
            return mu[0], s2[0]
        return mu[0]

    def _posterior_arrays(self, Q, getvar=True):
        Q = _lib.f64(np.atleast_2d(Q))
        M = len(Q)
        self._push_prior()
        mu = np.empty(M)
        s2 = np.empty(M) if getvar else None
        _lib.check(_lib.lib.ibo_posterior_batch(self._handle(), M, _lib.dp(Q), _lib.CLAMP_PY, _lib.dp(mu),
                                                _lib.dp(s2) if getvar else None))
        if getvar and self._augdev is not None:
            # variance from the augmented factor (addObservationPoint), :214-223
            mu2 = np.empty(M)
            _lib.check(_lib.lib.ibo_posterior_batch(self._augdev.h, M, _lib.dp(Q), _lib.CLAMP_PY, _lib.dp(mu2),
                                                    _lib.dp(s2)))
        return mu, s2

    def posteriors(self, X):
        """arrays of posterior means and variances (:231-244) -- one batched GPU call"""
        if len(self.X) == 0:
            pairs = [self.posterior(x) for x in X]
            return np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
        Q = np.array([np.atleast_1d(np.asarray(x, dtype=float)) for x in X])
        return self._posterior_arrays(Q, True)

    def mu(self, x):
        return self.posterior(x, getvar=False)

    def negmu(self, x):
        return -self.mu(x)

    EXTEND_MAX = 16          # more points than this in one addData call: a refit is cheaper than point-by-point rows

    def addData(self, X, Y, G=None):
        """append observations (:267-308).  X (N,D) or (D,); Y vector or scalar.  A fitted model is extended
        as the reference does -- z = L^-1 m, d = chol(r - z^T z), here with W = L^-1 on the device (ibo_gp_extend,
        O(N^2) per point); the first batch, large batches and a model whose row padding is full are (re)fitted."""
        if G is not None:
            raise NotImplementedError("gradient observations are not supported")
        X = np.array(X, dtype=float, ndmin=2)
        Y = np.array(Y, dtype=float, ndmin=1).flatten()
        assert len(Y) == len(X), 'wrong number of Y-observations given'
        if len(self.X) == 0 and len(self.gnoise) == 1:
            self.gnoise = np.tile(self.gnoise, X.shape[1])
        oldX, oldY = self.X, self.Y
        if len(self.X) == 0:
            self.X = np.copy(X)
            self.Y = np.copy(Y)
        else:
            self.X = np.r_[self.X, X]
            self.Y = np.r_[self.Y, Y]
        try:
            if not self._extend_device(len(oldX), X):
                self._fit_device()
        except Exception:
            # as the reference: a failed factorisation leaves the model as it was.  The device rows may have been
            # half written, so the old data are factored again before the error is passed on
            self.X, self.Y = oldX, oldY
            if len(oldX):
                self._fit_device()
            raise

    def _extend_device(self, n_old, Xnew):
        """block extension on the device; False when the handle has to be refitted instead"""
        if n_old == 0 or self._dev is None or len(Xnew) > self.EXTEND_MAX:
            return False
        spec = self.kernel._ibo_spec()
        if getattr(self, "_fit_spec", None) != (spec[0], tuple(spec[1]), spec[2], float(self.noise)):
            return False                             # kernel or noise changed since the fit
        info = ctypes.c_int(0)
        Xc = _lib.f64(Xnew); Yc = _lib.f64(self.Y)
        rc = _lib.lib.ibo_gp_extend(self._handle(), len(Xnew), _lib.dp(Xc), _lib.dp(Yc), ctypes.byref(info))
        if rc == _lib.ERR_STATE:
            return False
        _lib.check(rc)
        self._cache = {}
        return True

    def getYfromX(self, qx):
        for x, y in zip(self.X, self.Y):
            if np.all(qx == x):
                return y
        return None

    def done(self, x):
        self.selected = x
        self.endtime = time()

    def __deepcopy__(self, memo):
        from copy import deepcopy
        g = GaussianProcess(deepcopy(self.kernel, memo), prior=self.prior, noise=self.noise, device=self._device)
        if len(self.X):
            g.addData(self.X.copy(), self.Y.copy())
        return g


class PrefGaussianProcess(GaussianProcess):
    """GP trained on pairwise preferences (ego/gaussianprocess/__init__.py:331-527)."""

    def __init__(self, kernel, prefs=None, **kwargs):
        super(PrefGaussianProcess, self).__init__(kernel, **kwargs)
        self.preferences = []
        self.C = None
        if prefs is not None:
            self.addPreferences(prefs)

    # GP.C (ego/gaussianprocess/__init__.py:476-486) is a public attribute, but nothing on the device path reads the dense N x N matrix: the
    # device forms C, C^-1 and R + C^-1 from the pairs' entries (ibo_pref_finish).  It is built on the host when someone asks for it
    # (np.eye(n) * 5 and four np.add.at passes were a quarter of an addPreferences call at 512 pairs).
    @property
    def C(self):
        if self._C is None and self._C_spec is not None:
            n, pv, pu, w, reg = self._C_spec
            C = np.eye(n) * 5
            np.add.at(C, (pv, pu), -w); np.add.at(C, (pu, pv), -w)
            np.add.at(C, (pv, pv), w); np.add.at(C, (pu, pu), w)
            for _ in range(reg):                  # the reference's regulariser, one identity at a time (:491-496)
                C += np.eye(n)
            self._C = C
        return self._C

    @C.setter
    def C(self, value):
        self._C = value
        self._C_spec = None

    @staticmethod
    def _S_terms(y, Ry, v, u, w, hess=True):
        """MAP functional S(y) = -sum (d+1) log(Phi((y_v-y_u)/sqrt 2) + 1e-10) + y^T R^-1 y / 2
        (:351-385) with gradient and the per-pair Hessian weights; Ry = R^-1 y.  Phi is the reference's CDF
        (NR erf, truncated 1/sqrt 2) so S is the same function; its derivative uses the exact pdf."""
        z = (y[v] - y[u]) / np.sqrt(2)
        cdf = CDF(z) + 1e-10
        t = z * 0.707106
        pdf = 0.707106 / np.sqrt(np.pi) * np.exp(-t * t)           # d CDF / dz
        S = -np.sum(w * np.log(cdf)) + 0.5 * y.dot(Ry)
        ratio = pdf / cdf
        gz = -w * ratio / np.sqrt(2)
        g = Ry.copy()
        np.add.at(g, v, gz)
        np.add.at(g, u, -gz)
        if not hess:
            return S, g, None
        # -d/dz (pdf/cdf) = 2 t' pdf/cdf * 0.707106 + (pdf/cdf)^2, t' = 0.707106 z; (dz/dy)^2 = 1/2
        rho = w * (2 * 0.707106 * t * ratio + ratio * ratio) / 2.0
        return S, g, rho

    @staticmethod
    def _pair_sum_entries(n, v, u, w):
        """the distinct entries of sum_p w_p (e_v - e_u)(e_v - e_u)^T as (row * n + col, value), each summed in the
        order the reference's four scatter-adds visit it (the (v,v) terms, then (u,u), (v,u), (u,v))"""
        lin = np.concatenate([v * n + v, u * n + u, v * n + u, u * n + v]).astype(np.int64)
        val = np.concatenate([w, w, -w, -w])
        uniq, inv = np.unique(lin, return_inverse=True)
        acc = np.zeros(len(uniq))
        np.add.at(acc, inv, val)
        return np.ascontiguousarray(uniq, dtype=np.int64), np.ascontiguousarray(acc)

    def _map_newton(self, start, prefinds, tol=1e-9, maxit=100):
        """minimise the (convex) MAP functional by damped Newton.  Each step solves (R^-1 + C_pref) delta = -g on
        the GPU, where R^-1 and the Hessian live (ibo_pref_newton_step): vectors and the pairs' matrix entries
        travel, nothing N x N does.  The line search needs R^-1 (y + t delta) = R^-1 y + t R^-1 delta only."""
        import ctypes
        h = self._handle()
        v = np.array([i[0] for i in prefinds]); u = np.array([i[1] for i in prefinds])
        w = np.array([i[2] + 1.0 for i in prefinds])
        y = _lib.f64(np.array(start, dtype=float))
        N = len(y)
        Ry = np.empty(N)
        _lib.check(_lib.lib.ibo_pref_rinv_mul(h, _lib.dp(y), _lib.dp(Ry)))
        S, g, rho = self._S_terms(y, Ry, v, u, w)
        info = ctypes.c_int(0)
        for it in range(maxit):
            if np.max(np.abs(g)) < tol * max(1.0, np.max(np.abs(y))):
                break
            lin, val = self._pair_sum_entries(N, v, u, rho)
            delta = np.empty(N); Rdelta = np.empty(N)
            gc = _lib.f64(g)
            _lib.check(_lib.lib.ibo_pref_newton_step(h, len(lin), lin.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _lib.dp(val),
                                                     _lib.dp(gc), _lib.dp(delta), _lib.dp(Rdelta), ctypes.byref(info)))
            step = 1.0
            while True:
                Sn, gn, rn = self._S_terms(y + step * delta, Ry + step * Rdelta, v, u, w)
                if np.isfinite(Sn) and Sn <= S + 1e-4 * step * g.dot(delta):
                    break
                step *= 0.5
                if step < 1e-10:
                    return y
            y = y + step * delta
            Ry = Ry + step * Rdelta
            S, g, rho = Sn, gn, rn
            if it % 8 == 7:                           # R^-1 y afresh now and then: the running sum drifts by rounding only
                _lib.check(_lib.lib.ibo_pref_rinv_mul(h, _lib.dp(_lib.f64(y)), _lib.dp(Ry)))
                S, g, rho = self._S_terms(y, Ry, v, u, w)
        return y

    @staticmethod
    def _point_key(x):
        """hashable identity of a point: its fp64 bytes (+0.0 folds -0.0 into 0.0, as == would)"""
        return (np.asarray(x, dtype=np.float64) + 0.0).tobytes()

    @classmethod
    def _index_preferences(cls, prefs):
        """distinct points of all (preferred, unpreferred, degree) triples in order of first appearance
        (preferred before unpreferred within a triple; the reference's numbering, :395-406, which callers
        see through GP.X) -> (points (n, D), [(i_preferred, i_unpreferred, degree)], was-ever-preferred mask)"""
        if len(prefs) == 0:
            return np.zeros((0, 0)), [], []
        # all 2P points at once: identity of a point = its fp64 bytes (+0.0 folds -0.0 into 0.0, as == would)
        A = np.array([np.asarray(x, dtype=np.float64) for tr in prefs for x in (tr[0], tr[1])], dtype=np.float64)
        A = A.reshape(len(A), -1)
        K = np.ascontiguousarray(A + 0.0)
        keys = K.view(np.dtype((np.void, K.dtype.itemsize * K.shape[1]))).ravel()
        _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
        order = np.argsort(first, kind="stable")               # distinct points by first appearance
        number = np.empty(len(order), dtype=np.int64)
        number[order] = np.arange(len(order))
        k = number[np.asarray(inv).ravel()]
        fav = np.zeros(len(order), dtype=bool)
        fav[k[0::2]] = True
        pairs = [(int(i), int(j), tr[2]) for i, j, tr in zip(k[0::2], k[1::2], prefs)]
        return A[first[order]], pairs, fav.tolist()

    def addPreferences(self, prefs, useC=True, showPrefLikelihood=False):
        """add (x_preferred, x_unpreferred, degree) triples and refit from ALL preferences (:347-498)"""
        self.preferences.extend(prefs)
        newX, prefinds, preferred = self._index_preferences(self.preferences)
        # warm start of the MAP (:408-427): a point keeps the latent value it had; a new point starts at the
        # top of the current range if it was ever preferred, at the bottom otherwise
        top, bottom = (max(self.Y), min(self.Y)) if len(self.Y) > 0 else (.5, -.5)
        if len(self.Y) > 0:
            had = dict((self._point_key(x), y) for x, y in zip(self.X, self.Y))
            start = [had.get(self._point_key(x), top if fav else bottom) for x, fav in zip(newX, preferred)]
        else:
            start = [top if fav else bottom for fav in preferred]

        # K(X,X), Cholesky and L^-1 on the GPU (:432-438)
        self.X = newX
        self.Y = np.array(start, dtype=float)
        self.C = None
        self._augdev = None
        self.augR = self.augL = self.augX = None
        self._fit_device()
        _lib.check(_lib.lib.ibo_pref_begin(self._handle()))       # R^-1 = W^T W, on the device

        # MAP (:442).  The reference runs fmin_bfgs with numerical gradients (gtol 1e-5) on this
        # convex functional; Newton with the analytic Hessian reaches the same optimum in ~10 solves.
        Y = self._map_newton(start, prefinds)
        unpreferred = set(c for _, c, _ in prefinds)
        for r, c, _ in prefinds:                      # order fix-up (:445-457)
            if Y[r] <= Y[c]:
                if r not in unpreferred:
                    Y[r] = Y[c] + .1
        self._set_map(Y, prefinds, plain_fitted=True)

    def _inv_spd(self, A):
        """inverse of the (symmetric positive-definite) C matrix on the GPU"""
        A = _lib.f64(A)
        out = np.empty_like(A)
        _lib.check(_lib.lib.ibo_spd_inverse(self._dev.device, len(A), _lib.dp(A), _lib.dp(out), None))
        return out

    def _set_map(self, Y, prefinds, plain_fitted=False):
        """everything downstream of the MAP (:459-498): C matrix, L = chol(R + C^-1).
        The C-matrix loop reads mu with L = chol(R) (:476), so the device must hold the
        plain factorisation at this point."""
        self.Y = np.array(Y, dtype=float)
        if not plain_fitted:
            self.C = None
            self._fit_device()
        _lib.check(_lib.lib.ibo_gp_set_y(self._handle(), _lib.dp(_lib.f64(self.Y))))
        mu = self._posterior_arrays(self.X, getvar=False)[0]      # L = chol(R) at this point
        n = len(self.X)
        # one weight per pair (all pairs at once), scattered into the four entries it touches; np.add.at keeps the
        # reference's accumulation for points that occur in several pairs
        pv = np.array([p[0] for p in prefinds], dtype=int); pu = np.array([p[1] for p in prefinds], dtype=int)
        d = (mu[pv] - mu[pu]) / (np.sqrt(2) * np.sqrt(self.noise))
        cdf = np.maximum(CDF(d), 1e-10)
        pdf = np.maximum(PDF(d), 1e-10)
        w = 1.0 / (2 * self.noise) * (pdf ** 2 / cdf ** 2 + d * pdf / cdf)
        self.C = None
        self._C_spec = (n, pv, pu, w, 0)             # GP.C, formed when read (the property above)
        # L = chol(R + C^-1) (:488-497): C, its inverse and the sum are formed on the device from the pairs' entries
        # (ibo_pref_finish); a C too ill-conditioned to factor gets the reference's regulariser, one identity at a time
        import ctypes
        if not plain_fitted:                         # (addPreferences has done this before its Newton steps)
            _lib.check(_lib.lib.ibo_pref_begin(self._handle()))
        lin, val = self._pair_sum_entries(n, pv, pu, w)
        info = ctypes.c_int(0)
        for i in range(11):
            rc = _lib.lib.ibo_pref_finish(self._handle(), len(lin), lin.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _lib.dp(val),
                                          5.0 + i, ctypes.byref(info))
            if rc == _lib.ERR_NOT_PD:
                print('[addPreferences] GP.C matrix is ill-conditioned, adding regularizer delta = %d' % (i + 1))
                self.C = None
                self._C_spec = (n, pv, pu, w, i + 1)
                continue
            _lib.check(rc)
            break
        else:
            raise NotPositiveDefinite(_lib.ERR_NOT_PD, "R + C^-1 could not be factored")
        self._cache = {}
        self._prior_pushed = False
        self._push_prior()
        self._fit_spec = None

    def addObservationPoint(self, X):
        """add a point to observe at, without its observation (:502-519)"""
        X = np.array(X, dtype=float, ndmin=2)
        if self.augR is None:
            self.augR = self.R.copy()
            self.augX = self.X.copy()
        ktype, hyper, sf2, _ = self.kernel._ibo_spec()
        augX = np.r_[self.augX, X]
        n = len(augX)
        K = np.empty((n, n))
        Xc = _lib.f64(augX)
        _lib.check(_lib.lib.ibo_cov_matrix(self._dev.device, ktype, Xc.shape[1], _lib.dp(hyper), len(hyper), sf2, n,
                                           _lib.dp(Xc), 0, None, _lib.DIAG_UNIT_PLUS_NOISE, float(self.noise),
                                           _lib.dp(K)))
        self.augR = K
        invC = np.zeros_like(K)
        m = self.C.shape[0]
        invC[:m, :m] = self._inv_spd(self.C)
        self.augX = augX
        if self._augdev is None:
            self._augdev = _DeviceGP(self._dev.device)
        self._fit_device(A=K + invC, dev=self._augdev, X=augX)
        Lh = np.empty((n, n))
        _lib.check(_lib.lib.ibo_gp_get_L(self._augdev.h, _lib.dp(Lh)))
        self.augL = Lh

    def addData(self, X, Y, G=None):
        if getattr(self, "preferences", None) is None:      # called from the base constructor
            return GaussianProcess.addData(self, X, Y, G)
        raise NotImplementedError("can't (yet) add explicit ratings to preference GP")
