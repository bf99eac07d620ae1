"""
Marginal likelihood of the GP hyper-parameters (ego/gaussianprocess/trainhyper.py).

    marginalLikelihood(kernel, X, Y, nhyper, computeGradient=True, useCholesky=True, noise=1e-3)
    nlml(loghyper, Kernel, X, Y)     dnlml(loghyper, Kernel, X, Y)     nlmlMulti(...)
    nlml_grid(KernelClass, thetas, X, Y, noise=1e-3)      <- new: a whole theta grid in one call

The value (K assembly, Cholesky, L^-1 Y, log det) is computed on the GPU by
ibo_nlml_grid; value + gradient for one theta by ibo_nlml_grad (K^-1 = W^T W from the device
factorisation, then one fused kernel contracts K^-1 - alpha alpha^T with every dK/dtheta_i,
recomputed from X on the fly).
"""
import numpy as np
from numpy.linalg import LinAlgError

from .. import _lib


def _spec_rows(kernels):
    specs = [k._ibo_spec() for k in kernels]
    ktype = specs[0][0]
    thetas = _lib.f64(np.array([s[1] for s in specs]))
    sf2 = _lib.f64([s[2] for s in specs])
    return ktype, thetas, sf2


def nlml_values(kernels, X, Y, noise=1e-3, device=None, spec=None):
    """NLML for a list of kernel objects of one class (a theta grid); NaN where K is not PD.
    spec: (ktype, rows, sf2) as Kernel._ibo_spec_rows gives them, instead of the objects."""
    X = _lib.rows(X); Y = _lib.f64(Y)
    N, D = X.shape
    ktype, thetas, sf2 = _spec_rows(kernels) if spec is None else spec
    out = np.empty(len(thetas))
    dev = _lib.default_device() if device is None else device
    _lib.check(_lib.lib.ibo_nlml_grid(dev, ktype, N, D, _lib.dp(X), _lib.dp(Y), len(thetas), _lib.dp(thetas),
                                      thetas.shape[1], _lib.dp(sf2), float(noise), _lib.dp(out)))
    return out


def nlml_grid(Kernel, thetas, X, Y, noise=1e-3, device=None):
    """NLML at every row of `thetas` (hyper-parameters, NOT logs); returns (values, argmin)."""
    vals = nlml_values(None, X, Y, noise, device, spec=Kernel._ibo_spec_rows(thetas))
    return vals, int(np.nanargmin(vals))


def marginalLikelihood(kernel, X, Y, nhyper, computeGradient=True, useCholesky=True, noise=1e-3):
    """negative log marginal likelihood and (optionally) its partial derivatives w.r.t.
    each log hyper-parameter (trainhyper.py:47-95).  useCholesky is accepted for
    signature compatibility; the device path always factors."""
    NX = len(X)
    assert NX == len(Y)
    if not computeGradient:
        v = nlml_values([kernel], X, Y, noise)[0]
        if not np.isfinite(v):
            raise LinAlgError("covariance matrix is not positive definite for hyperparameters %s"
                              % (kernel.hyperparams,))
        return v
    # value and gradient in one device call: Cholesky, L^-1, K^-1 = W^T W, then
    # dnlml_i = sum((K^-1 - alpha alpha^T) * dK/dtheta_i) / 2   (:70-71)
    import ctypes
    Xa = _lib.rows(X); Ya = _lib.f64(Y)
    N, D = Xa.shape
    ktype, hyper, sf2, _ = kernel._ibo_spec()
    spec = kernel._ibo_grad_spec(D)[:nhyper]
    if len(spec) < nhyper:
        raise ValueError("kernel has %d hyperparameters, %d gradients requested" % (len(spec), nhyper))
    modes = (ctypes.c_int * nhyper)(*[m for m, _ in spec])
    dims = (ctypes.c_int * nhyper)(*[d for _, d in spec])
    v = ctypes.c_double()
    g = np.empty(nhyper)
    try:
        _lib.check(_lib.lib.ibo_nlml_grad(_lib.default_device(), ktype, N, D, _lib.dp(Xa), _lib.dp(Ya), _lib.dp(hyper),
                                          len(hyper), sf2, float(noise), nhyper, modes, dims, ctypes.byref(v), _lib.dp(g)))
    except _lib.NotPositiveDefinite:
        raise LinAlgError("covariance matrix is not positive definite for hyperparameters %s" % (kernel.hyperparams,))
    return v.value, g


# fmin_bfgs asks for f(x) and then f'(x) at the same x (also inside its line search); both come out of one
# factorisation, so the pair is computed once and the last one is remembered
_last = {"key": None, "value": None, "grad": None}


def _value_and_grad(loghyper, kernel, X, Y):
    loghyper = np.asarray(loghyper, dtype=float)
    # keyed on the CONTENT of the data (a digest: 0.1 ms at N=4096, D=16 against a 9 ms factorisation), so an
    # in-place edit of X or Y can never return the pair computed for the old data
    import hashlib
    dig = hashlib.blake2b(_lib.rows(X).tobytes(), digest_size=16)
    dig.update(_lib.f64(Y).tobytes())
    key = (loghyper.tobytes(), kernel, dig.digest())
    if _last["key"] != key:
        k = kernel(np.exp(loghyper))
        _last["key"] = None
        _last["value"], _last["grad"] = marginalLikelihood(k, X, Y, len(loghyper), computeGradient=True)
        _last["key"] = key
    return _last["value"], _last["grad"]


def nlml(loghyper, kernel, X, Y, *args):
    """NLML as a function of LOG hyper-parameters (trainhyper.py:99-115); 100 when not PD."""
    try:
        ml = _value_and_grad(loghyper, kernel, X, Y)[0]
    except LinAlgError as e:
        print(e)
        ml = 100
        print('returning nlml = 100')
    return ml


def nlmlMulti(loghyper, kernel, X, Y, *args):
    k = kernel(np.exp(loghyper))
    ml = 0.0
    for x, y in zip(X, Y):
        ml += marginalLikelihood(k, x, y, len(loghyper))[0]
    return ml


def dnlml(loghyper, kernel, X, Y):
    return _value_and_grad(loghyper, kernel, X, Y)[1]
