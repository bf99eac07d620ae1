"""
Acquisition functions and their maximisers, with the reference's names and
defaults (ego/acquisition/__init__.py):

    EI(GP, xi=.01)  PI(GP, xi=.01)  UCB(GP, NA, delta=0.1, scale=0.2)    .negf(x) / .f(x)
    maximizeEI(model, bounds, useCDIRECT=True, xi=0.01, maxiter=50, maxtime=30, maxsample=10000)
    maximizePI(model, bounds, xi=0.01, maxiter=50, maxtime=30, maxsample=10000, useCDIRECT=True)
    maximizeUCB(model, bounds, delta=0.1, scale=0.2, useCDIRECT=True, maxiter=50, maxtime=30, maxsample=10000)
        -> (opt, optx)

plus the batched entry point the GPU makes worthwhile:

    sweep(model, candidates, acq='ei', ...) -> dict(best_val, best_idx, [mu, s2, acq])

The default maximize* path is the reference's cdirectGP -> acqmaxGP -> DIRECT
(ego/acquisition/__init__.py:307-447, cpp/optimizeGP.cpp:262-349) with the tree
on the host and every batch of sample points evaluated by the HIP sweep kernel.
"""
import ctypes

import numpy as np

from .. import _lib
from ..gaussianprocess import GaussianProcess, PrefGaussianProcess, CDF, PDF      # noqa: F401
from ..utils.optimize import direct, cdirect
from ..utils.latinhypercube import lhcSample                                        # noqa: F401

_ACQ = {'ei': _lib.ACQ_EI, 'pi': _lib.ACQ_PI, 'ucb': _lib.ACQ_UCB}


class UCB(object):
    """upper confidence bound; note sqrt(scale * sBeta) with sBeta already a square
    root -- the class and the native path disagree in the reference and both are
    kept (ego/acquisition/__init__.py:60-71 vs :319)"""

    def __init__(self, GP, NA, delta=0.1, scale=0.2, **kwargs):
        super(UCB, self).__init__()
        self.GP = GP
        self.scale = scale
        t = len(self.GP.Y) + 1
        self.sBeta = np.sqrt(2.0 * np.log(t ** (NA // 2 + 2) * np.pi ** 2 / (3.0 * delta)))

    def negf(self, x):
        r = self.GP._eval(x, _lib.ACQ_UCB, np.sqrt(self.scale * self.sBeta), _lib.ERF_NR, _lib.CLAMP_PY, ("acq",))
        return -r["acq"][0]

    def f(self, x):
        return -self.negf(x)


class PI(object):
    """probability of improvement (:100-114); NR-erf CDF as the Python reference"""

    def __init__(self, GP, xi=.01, **kwargs):
        super(PI, self).__init__()
        self.GP = GP
        self.xi = xi
        self.Z = np.max(self.GP.Y) + xi

    def negf(self, x):
        r = self.GP._eval(x, _lib.ACQ_PI, self.xi, _lib.ERF_NR, _lib.CLAMP_PY, ("acq",), ymax=self.Z - self.xi)
        return -r["acq"][0]

    def f(self, x):
        return -self.negf(x)


class EI(object):
    """expected improvement (:138-169); NR-erf CDF/PDF as the Python reference"""

    def __init__(self, GP, xi=.01, **kwargs):
        super(EI, self).__init__()
        self.GP = GP
        self.ymax = np.max(self.GP.Y)
        self.xi = xi
        assert np.isscalar(self.ymax)
        assert np.isscalar(self.xi)

    def negf(self, x):
        r = self.GP._eval(x, _lib.ACQ_EI, self.xi, _lib.ERF_NR, _lib.CLAMP_PY, ("acq",), ymax=self.ymax)
        return -r["acq"][0]

    def f(self, x):
        return -self.negf(x)

    def values(self, X):
        """EI at many points at once (one GPU launch)"""
        return self.GP._eval(X, _lib.ACQ_EI, self.xi, _lib.ERF_NR, _lib.CLAMP_PY, ("acq",), ymax=self.ymax)["acq"]


def _ucb_parm(model, bounds, delta, scale):
    """sigma multiplier of the native UCB (ego/acquisition/__init__.py:316-319;
    NA/2 there is Python-2 integer division)"""
    t = len(model.Y) + 1
    NA = len(bounds)
    return float(np.sqrt(scale * 2.0 * np.log(t ** (NA // 2 + 2) * np.pi ** 2 / (3.0 * delta))))


def cdirectGP(model, bounds, maxiter, maxtime, maxsample, acqfunc=None, xi=-1, beta=-1, scale=-1, delta=-1, compat=True,
              return_samples=False, **kwargs):
    """cdirectGP (ego/acquisition/__init__.py:307-468), same name, positional order and defaults (`beta` is accepted and
    unused, as in the reference): same enum mapping and `parm`, then DIRECT on the GPU objective.  The model is already
    factored on the device, so the per-call linalg.inv(R) of the reference disappears.  `compat` / `return_samples` are
    additions (the dimension-0 stall switch, the sample count)."""
    if acqfunc == 'ei' or acqfunc == 'pi':
        parm = xi
    elif acqfunc == 'ucb':
        parm = _ucb_parm(model, bounds, delta, scale)
    else:
        raise NotImplementedError('unknown acquisition function %s' % acqfunc)
    if len(model.X) == 0:
        raise ValueError("model has no data")
    _, _, sf2_py, sf2_native = model.kernel._ibo_spec()
    lb = _lib.f64([b[0] for b in bounds]); ub = _lib.f64([b[1] for b in bounds])
    D = len(lb)
    opt = ctypes.c_double(); optx = np.empty(D); ns = ctypes.c_int64()
    h = model._handle()
    model._push_prior()
    _lib.check(_lib.lib.ibo_gp_set_kstar_sf2(h, sf2_native))
    try:
        _lib.check(_lib.lib.ibo_direct_max(h, D, _lib.dp(lb), _lib.dp(ub), _ACQ[acqfunc], float(parm),
                                           _lib.ERF_LIBM, _lib.CLAMP_NATIVE, int(maxiter), int(maxtime),
                                           int(maxsample), 1 if compat else 0, ctypes.byref(opt), _lib.dp(optx),
                                           ctypes.byref(ns)))
    finally:
        _lib.check(_lib.lib.ibo_gp_set_kstar_sf2(h, sf2_py))
    if return_samples:
        return opt.value, optx, ns.value
    return opt.value, optx


gpuDirectGP = cdirectGP          # the name earlier rounds of this package used


def maximizeUCB(model, bounds, delta=0.1, scale=0.2, useCDIRECT=True, maxiter=50, maxtime=30, maxsample=10000,
                **kwargs):
    """maximise the GP-UCB of [Srinivas 2009] (:78-96)"""
    if not useCDIRECT:
        print('using DIRECT')
        ucb = UCB(model, len(bounds), delta=delta, scale=scale, **kwargs)
        # the reference forgets to forward the budgets here (:86) and raises; they are forwarded
        opt, optx = direct(ucb.negf, bounds, maxiter=maxiter, maxtime=maxtime, maxsample=maxsample)
        return -opt, optx
    if isinstance(model, GaussianProcess):
        return cdirectGP(model, bounds, maxiter, maxtime, maxsample, acqfunc='ucb', delta=delta, scale=scale,
                           **kwargs)
    raise ValueError


def maximizePI(model, bounds, xi=0.01, maxiter=50, maxtime=30, maxsample=10000, useCDIRECT=True, **kwargs):
    """maximise the probability of improvement [Lizotte 2008] (:117-134)"""
    if not useCDIRECT:
        print('using DIRECT')
        pi = PI(model, xi, **kwargs)
        opt, optx = direct(pi.negf, bounds, maxiter=maxiter, maxtime=maxtime, maxsample=maxsample)
        return -opt, optx
    if isinstance(model, GaussianProcess):
        return cdirectGP(model, bounds, maxiter, maxtime, maxsample, acqfunc='pi', xi=xi, **kwargs)
    raise ValueError


def maximizeEI(model, bounds, useCDIRECT=True, xi=0.01, maxiter=50, maxtime=30, maxsample=10000, **kwargs):
    """maximise expected improvement (:174-197)"""
    if not useCDIRECT:
        print('using DIRECT')
        ei = EI(model, xi, **kwargs)
        opt, optx = direct(ei.negf, bounds, maxiter=maxiter, maxtime=maxtime, maxsample=maxsample)
        return -opt, optx
    if isinstance(model, GaussianProcess):
        return cdirectGP(model, bounds, maxiter, maxtime, maxsample, acqfunc='ei', xi=xi, **kwargs)
    raise ValueError


def sweep(model, candidates, acq='ei', xi=0.01, delta=0.1, scale=0.2, parm=None, native=True, ymax=None,
          exclude=None, exclude_radius=0.5, index_base=0, outputs=(), NA=None, incremental=False, exchange=None):
    """Evaluate an acquisition over a whole candidate array and return its arg-max.

    candidates   (M, D) ndarray (uploaded) or a _lib.DeviceArray already in HBM
    native=True  libego semantics: libm erf, variance clamp [1e-8, 10], k* with libego's sf2
    native=False Python-class semantics: NR erf, clamp [1e-7, 10]
    exclude      points whose exclude_radius-ball is left out of the arg-max (gallery rule)
    outputs      any of 'mu', 's2', 'acq': per-candidate arrays to return (host ndarrays)
    incremental  keep the per-candidate state of THIS DeviceArray on the model's handle and, when the model has only
                 grown through addData since the last such call, fold the new rows in instead of sweeping again
                 (fastUCBGallery's rounds).  Honoured for a caller-owned DeviceArray only (an ndarray is uploaded to a
                 temporary and swept in full); the state is keyed on the array's generation (_lib.DeviceArray.generation),
                 so a freed-and-reallocated or re-uploaded array is swept in full again
    exchange     an ibo_amd.multigpu.RcclArgmax: the sharded step in one device-side call (ibo_acq_sweep_exchange) -- this rank's
                 arg-max goes from the sweep's output words into the all-reduce buffer without visiting the host; the result then
                 also carries global_val, global_idx, global_x, global_rank (identical on every rank).  No `outputs` with it.
    Returns dict(best_val, best_idx, kernel_ms, [mu], [s2], [acq]); first maximiser wins ties.
    """
    if isinstance(candidates, _lib.DeviceArray):
        cand = candidates
    else:
        cand = _lib.DeviceArray.from_host(np.atleast_2d(candidates), model._dev.device)
        incremental = False
    M = cand.shape[0]
    code = _ACQ[acq]
    if parm is None:
        if acq == 'ucb':
            nd = NA if NA is not None else cand.shape[1]
            parm = _ucb_parm(model, [None] * nd, delta, scale) if native else \
                float(np.sqrt(scale * UCB(model, nd, delta, scale).sBeta))
        else:
            parm = xi
    _, _, sf2_py, sf2_native = model.kernel._ibo_spec()
    h = model._handle()
    model._push_prior()
    outs = {k: _lib.DeviceArray((M,), model._dev.device) for k in outputs}
    ex = None if exclude is None or len(exclude) == 0 else _lib.f64(np.atleast_2d(exclude))
    bv = ctypes.c_double(); bi = ctypes.c_int64()
    if native:
        _lib.check(_lib.lib.ibo_gp_set_kstar_sf2(h, sf2_native))
    glob = None
    try:
        if exchange is not None:
            if outputs:
                raise ValueError("per-candidate outputs are not available together with exchange=")
            gv = ctypes.c_double(); gi = ctypes.c_int64(); gr = ctypes.c_int(); gx = np.zeros(cand.shape[1])
            _lib.check(_lib.lib.ibo_acq_sweep_exchange(
                h, exchange.h, 1 if incremental else 0, M, cand.ptr, code, float(parm), _lib.ERF_LIBM if native else _lib.ERF_NR,
                _lib.CLAMP_NATIVE if native else _lib.CLAMP_PY, float('nan') if ymax is None else float(ymax),
                0 if ex is None else len(ex), None if ex is None else _lib.dp(ex), float(exclude_radius), int(index_base),
                ctypes.byref(bv), ctypes.byref(bi), ctypes.byref(gv), ctypes.byref(gi), _lib.dp(gx), ctypes.byref(gr)))
            glob = dict(global_val=gv.value, global_idx=gi.value, global_x=gx, global_rank=gr.value)
        entry = _lib.lib.ibo_acq_sweep_incremental if incremental else _lib.lib.ibo_acq_sweep
        if exchange is None:
            _lib.check(entry(
                h, M, cand.ptr, code, float(parm), _lib.ERF_LIBM if native else _lib.ERF_NR,
                _lib.CLAMP_NATIVE if native else _lib.CLAMP_PY, float('nan') if ymax is None else float(ymax),
                0 if ex is None else len(ex), None if ex is None else _lib.dp(ex), float(exclude_radius),
                int(index_base), outs["mu"].ptr if "mu" in outs else None, outs["s2"].ptr if "s2" in outs else None,
                outs["acq"].ptr if "acq" in outs else None, ctypes.byref(bv), ctypes.byref(bi)))
    finally:
        if native:
            _lib.check(_lib.lib.ibo_gp_set_kstar_sf2(h, sf2_py))
    ms = ctypes.c_float(); name = ctypes.c_char_p()
    _lib.check(_lib.lib.ibo_last_sweep_kernel_ms(h, ctypes.byref(ms), ctypes.byref(name)))
    res = dict(best_val=bv.value, best_idx=bi.value, kernel_ms=ms.value, kernel=name.value.decode())
    if glob is not None:
        res.update(glob)
    for k, v in outs.items():
        res[k] = v.to_host()
    return res
