"""
Covariance kernels -- same classes, constructor arguments, attributes and
methods as the reference's ego/gaussianprocess/kernel.py, written for Python 3.

`cov(x1, x2)` is the scalar host-side definition (API surface, used for spot
checks); every matrix the model needs (covMatrix, the GP's R) is assembled on
the GPU by libibo_hip (linalg.hip: cov_matrix_kernel).  `_ibo_spec()` is the
flattening the device understands: (kernel type code, length scales, sf2).
"""
import numpy as np

from .. import _lib


class Kernel(object):
    """base class (ego/gaussianprocess/kernel.py:27-58)"""

    def __init__(self, hyperparams):
        self._hyperparams = np.array(hyperparams, dtype=float)
        self._hyperparams.setflags(write=False)

    def getHyperparams(self):
        return self._hyperparams

    hyperparams = property(getHyperparams)

    def cov(self, x1, x2):
        raise NotImplementedError('kernel-derived class does not have cov method')

    # (ktype, hyper for the device, sf2 the Python model applies, sf2 libego applies)
    def _ibo_spec(self):
        raise NotImplementedError

    # the device description of a whole grid of hyper-parameter rows (ibo_nlml_grid): (ktype, rows, sf2 per row) -- what
    # [cls(t)._ibo_spec() for t in thetas] says, which this default does; a class whose rows need no object each overrides it
    @classmethod
    def _ibo_spec_rows(cls, thetas):
        specs = [cls(np.asarray(t, dtype=float))._ibo_spec() for t in thetas]
        return specs[0][0], _lib.f64(np.array([sp[1] for sp in specs])), _lib.f64([sp[2] for sp in specs])

    # device description of derivative(X, hp) for hp = 0..nhyper-1: list of (mode, dim), see ibo_nlml_grad
    def _ibo_grad_spec(self, D):
        raise NotImplementedError

    def covMatrix(self, X, device=None):
        """K[i,j] = cov(X[i], X[j]), diagonal included (kernel.py:46-53); GPU-assembled."""
        X = _lib.rows(X)
        n, D = X.shape
        ktype, hyper, sf2, _ = self._ibo_spec()
        K = np.empty((n, n))
        dev = _lib.default_device() if device is None else device
        _lib.check(_lib.lib.ibo_cov_matrix(dev, ktype, D, _lib.dp(hyper), len(hyper), sf2, n, _lib.dp(X), 0, None,
                                           _lib.DIAG_KERNEL_PLUS_NOISE, 0.0, _lib.dp(K)))
        return K

    def derivative(self, X, hp):
        raise NotImplementedError('kernel-derived class does not have derivative method')


class SVKernel(object):
    """signal-variance mix-in (kernel.py:60-68)"""

    def __init__(self, mag):
        self._magnitude = mag
        self._sf2 = np.exp(2.0 * np.log(self._magnitude))

    def covScale(self, k):
        return self._sf2 * k


def _sqdiff(X):
    X = _lib.rows(X)
    return X[:, None, :] - X[None, :, :]


class GaussianKernel_iso(Kernel):
    """isotropic squared exponential, hyperparams = [theta] (kernel.py:71-106)"""

    def __init__(self, hyperparams, **kwargs):
        super(GaussianKernel_iso, self).__init__(hyperparams)
        self._itheta2 = 1 / float(np.asarray(hyperparams, dtype=float)[0]) ** 2

    def cov(self, x1, x2):
        d = np.asarray(x1, dtype=float) - np.asarray(x2, dtype=float)
        return float(np.exp(-.5 * np.linalg.norm(d) ** 2 * self._itheta2))

    def _ibo_spec(self):
        return _lib.K_SE_ISO, _lib.f64([1.0 / np.sqrt(self._itheta2)]), 1.0, 1.0

    def _ibo_grad_spec(self, D):
        return [(1, 0)]

    def derivative(self, X, hp):
        if hp != 0:
            raise ValueError
        K = self.covMatrix(X)          # includes sf2 for the SV subclass, as in the reference
        C = np.sum(_sqdiff(X) ** 2, axis=2) * self._itheta2
        return K * C


class SVGaussianKernel_iso(SVKernel, GaussianKernel_iso):
    """hyperparams = [theta, magnitude] (kernel.py:109-127)"""

    def __init__(self, hyperparams, **kwargs):
        hyperparams = np.asarray(hyperparams, dtype=float)
        GaussianKernel_iso.__init__(self, hyperparams[:-1])
        SVKernel.__init__(self, hyperparams[-1])
        self._hyperparams = np.array(hyperparams)
        self._hyperparams.setflags(write=False)

    def cov(self, x1, x2):
        return self.covScale(GaussianKernel_iso.cov(self, x1, x2))

    def _ibo_spec(self):
        # libego sees kernel type 1 and ignores the magnitude (cpp/optimizeGP.cpp:303-310)
        return _lib.K_SE_ISO, _lib.f64([1.0 / np.sqrt(self._itheta2)]), float(self._sf2), 1.0

    def _ibo_grad_spec(self, D):
        return [(1, 0), (2, 0)]

    def derivative(self, X, hp):
        if hp == 0:
            return GaussianKernel_iso.derivative(self, X, hp)
        elif hp == 1:
            return 2.0 * self.covMatrix(X)


class GaussianKernel_ard(Kernel):
    """anisotropic squared exponential, one length scale per dimension (kernel.py:130-166)"""

    def __init__(self, hyperparams, **kwargs):
        super(GaussianKernel_ard, self).__init__(hyperparams)
        self._theta = np.clip(np.asarray(hyperparams, dtype=float), 1e-4, 1e4)
        self._itheta2 = np.array([1.0 / t ** 2 for t in self._theta])

    def cov(self, x1, x2):
        d = np.asarray(x1, dtype=float) - np.asarray(x2, dtype=float)
        return float(np.exp(-.5 * np.sum(self._itheta2 * d ** 2)))

    def _ibo_spec(self):
        return _lib.K_SE_ARD, _lib.f64(self._theta), 1.0, 1.0

    @classmethod
    def _ibo_spec_rows(cls, thetas):
        # (64 objects cost 0.6 ms -- 2 % of the N = 4096 grid they describe; the rows are the constructor's clipped length scales)
        if cls is not GaussianKernel_ard:
            return super(GaussianKernel_ard, cls)._ibo_spec_rows(thetas)
        th = np.clip(np.asarray(thetas, dtype=float), 1e-4, 1e4)
        return _lib.K_SE_ARD, _lib.f64(th), _lib.f64(np.ones(len(th)))

    def _ibo_grad_spec(self, D):
        return [(0, d) for d in range(D)]

    def derivative(self, X, hp):
        NA = _lib.rows(X).shape[1]
        if not hp < NA:
            raise ValueError
        K = self.covMatrix(X)
        C = self._itheta2[hp] * _sqdiff(X)[:, :, hp] ** 2.0
        return K * C


class SVGaussianKernel_ard(SVKernel, GaussianKernel_ard):
    """hyperparams = [theta_1..theta_D, magnitude] (kernel.py:169-188)"""

    def __init__(self, hyperparams, **kwargs):
        hyperparams = np.asarray(hyperparams, dtype=float)
        GaussianKernel_ard.__init__(self, hyperparams[:-1])
        SVKernel.__init__(self, hyperparams[-1])
        self._hyperparams = np.array(hyperparams)
        self._hyperparams.setflags(write=False)

    def cov(self, x1, x2):
        return self.covScale(GaussianKernel_ard.cov(self, x1, x2))

    def _ibo_spec(self):
        return _lib.K_SE_ARD, _lib.f64(self._theta), float(self._sf2), 1.0

    def _ibo_grad_spec(self, D):
        return [(0, d) for d in range(D)] + [(2, 0)]

    def derivative(self, X, hp):
        if hp < len(self._theta):
            return GaussianKernel_ard.derivative(self, X, hp)
        elif hp == len(self._theta):
            return 2.0 * self.covMatrix(X)


class MaternKernel3(Kernel):
    """Matern nu=3/2, hyperparams = [theta, magnitude] (kernel.py:191-227)"""

    def __init__(self, hyperparams, **kwargs):
        super(MaternKernel3, self).__init__(hyperparams)
        self._theta = float(hyperparams[0])
        self._magnitude = float(hyperparams[1])
        self._sf2 = np.exp(2.0 * np.log(self._magnitude))
        self.sqrt3 = np.sqrt(3)

    def cov(self, x1, x2):
        d = np.asarray(x1, dtype=float) - np.asarray(x2, dtype=float)
        z = self.sqrt3 * np.linalg.norm(d) / self._theta
        return float(self._sf2 * (1.0 + z) * np.exp(-z))

    def _ibo_spec(self):
        # libego: sf2 = 1 for kernel type 2 whatever the magnitude (cpp/optimizeGP.cpp:303-310)
        return _lib.K_MATERN3, _lib.f64([self._theta]), float(self._sf2), 1.0

    def _ibo_grad_spec(self, D):
        return [(3, 0), (2, 0)]

    def derivative(self, X, hp):
        K = self.covMatrix(X)
        if hp == 0:
            r = np.sqrt(np.sum(_sqdiff(X) ** 2, axis=2))
            C = self._sf2 * r ** 2 * np.exp(-r)        # as the reference writes it (kernel.py:221)
            np.fill_diagonal(C, 0.0)
            return C
        elif hp == 1:
            return 2.0 * K
        raise ValueError


class MaternKernel5(Kernel):
    """Matern nu=5/2, hyperparams = [theta, magnitude] (kernel.py:230-266).
    The reference's cov() prints and returns None (kernel.py:246-249); the
    formula it computes -- and that its C++ twin uses, cpp/optimizeGP.cpp:108 --
    is what is returned here."""

    def __init__(self, hyperparams, **kwargs):
        super(MaternKernel5, self).__init__(hyperparams)
        self._theta = float(hyperparams[0])
        self._magnitude = float(hyperparams[1])
        self._sf2 = np.exp(2.0 * np.log(self._magnitude))

    def cov(self, x1, x2):
        d = np.asarray(x1, dtype=float) - np.asarray(x2, dtype=float)
        z = np.sum((np.sqrt(5.0) * d / self._theta) ** 2.0)
        return float(self._sf2 * np.exp(-np.sqrt(z)) * (1.0 + np.sqrt(z) + z / 3.0))

    def _ibo_spec(self):
        return _lib.K_MATERN5, _lib.f64([self._theta]), float(self._sf2), float(self._sf2)

    def _ibo_grad_spec(self, D):
        return [(4, 0), (2, 0)]

    def derivative(self, X, hp):
        K = self.covMatrix(X)
        if hp == 0:
            z = np.sum((np.sqrt(5.0) * _sqdiff(X) / self._theta) ** 2.0, axis=2)
            C = self._sf2 * (z + np.sqrt(z) ** 3.0) * np.exp(-np.sqrt(z)) / 3.0
            np.fill_diagonal(C, 0.0)
            return C
        elif hp == 1:
            return 2.0 * K
        raise ValueError
