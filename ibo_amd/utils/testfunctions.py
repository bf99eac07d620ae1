"""
Benchmark objectives for Bayesian-optimisation experiments, with the reference's class
names and attributes (ego/utils/testfunctions.py): `name`, `minimum`, `argmin`, `bounds`,
`maximize`, `f(x)`, `createKernel(KernelClass)`; plus `values(X)` for whole arrays.

Every analytic function is written once, vectorised over rows, as the quantity to be
MINIMISED; `f` flips the sign when `maximize` is set (the reference's default, because
the GP machinery maximises).  `Synthetic` draws a random function from a GP through the
device-backed GaussianProcess, and `learnHyper` fits a length scale by BFGS on the
device NLML.
"""
import numpy as np

from ..gaussianprocess import GaussianProcess
from ..gaussianprocess.kernel import GaussianKernel_iso, GaussianKernel_ard, MaternKernel3
from .latinhypercube import lhcSample


class TestFunction(object):
    __test__ = False            # not a pytest class

    def __init__(self, name, minimum, argmin, bounds, maximize=True, kernelHP=None, **kwargs):
        self.name = name
        self.maximize = maximize
        self.minimum = minimum
        self.argmin = argmin
        self.bounds = bounds
        self.defaultHP = kernelHP

    def _cost(self, X):
        """value to minimise at each row of X (n, D) -> (n,)"""
        raise NotImplementedError

    def values(self, X):
        c = self._cost(np.atleast_2d(np.asarray(X, dtype=float)))
        return -c if self.maximize else c

    def f(self, x):
        """f(x); negated when maximize=True (the default)"""
        return float(self.values(np.asarray(x, dtype=float).reshape(1, -1))[0])

    def createKernel(self, Kernel):
        """a kernel object of the given class with this function's tuned hyper-parameters"""
        if self.defaultHP is None or Kernel not in self.defaultHP:
            raise ValueError('test function %s has no default values for kernel %s' % (self.name, Kernel.__name__))
        return Kernel(np.array(self.defaultHP[Kernel]))


def _box(lo, hi, d):
    return np.array([[lo, hi]] * d, dtype=float)


# --------------------------------------------------------------------------- one-dimensional curves (testfunctions.py:74-124)
class _Curve(TestFunction):
    """a function of the first coordinate only"""
    hp = None

    def __init__(self, name, minimum, argmin, lo, hi, **kwargs):
        super(_Curve, self).__init__(name, minimum, argmin, _box(lo, hi, 1), **kwargs)
        self.defaultHP = {GaussianKernel_iso: [self.hp[0], 1], GaussianKernel_ard: [self.hp[1], 1], MaternKernel3: [self.hp[2], 1]}


class Poly4(_Curve):
    """|x^3 + x^2 + x| / 100 on [-10, 10] (named "4th-order" in the reference; the cubic is what it evaluates)"""
    hp = (1.620, 1.628, 4.635)

    def __init__(self, **kwargs):
        super(Poly4, self).__init__("4th-order poly", 0, [0], -10., 10., **kwargs)

    def _cost(self, X):
        t = X[:, 0]
        return np.abs(t ** 3 + t ** 2 + t) / 100.0


class Poly6(_Curve):
    """Goldstein's sixth-order polynomial ((x^2 - 15) x^2 + 27) x^2 + 250, over 100, on [-4, 4]"""
    hp = (0.285, 0.287, 0.678)

    def __init__(self, **kwargs):
        super(Poly6, self).__init__("Goldstein 6th-order", 0.07, [0], -4., 4., **kwargs)

    def _cost(self, X):
        q = X[:, 0] ** 2
        return (((q - 15) * q + 27) * q + 250) / 100


class Schubert1(_Curve):
    """sum_{i=1..5} i cos((i + 1) x + i) on [-1, 1]"""
    hp = (0.192, 0.192, 0.279)

    def __init__(self, **kwargs):
        super(Schubert1, self).__init__("Schubert", -8.5178, [-0.195], -1., 1., **kwargs)

    def _cost(self, X):
        i = np.arange(1, 6)
        return np.sum(i * np.cos((i + 1) * X[:, :1] + i), axis=1)


# --------------------------------------------------------------------------- Shekel family (testfunctions.py:150-207)
_SHEKEL_A = np.array([[4, 4, 4, 4], [1, 1, 1, 1], [8, 8, 8, 8], [6, 6, 6, 6], [3, 7, 3, 7], [2, 9, 2, 9],
                      [5, 5, 3, 3], [8, 1, 8, 1], [6, 2, 6, 2], [7, 3.6, 7, 3.6]], dtype=float)
_SHEKEL_C = np.array([.1, .2, .2, .4, .4, .6, .3, .7, .5, .5])


class Shekel(TestFunction):
    """-sum_i 1 / (|x - a_i|^2 + c_i) over the first m wells, x in [0, 10]^4"""
    wells = 10

    def __init__(self, name, minimum, argmin, **kwargs):
        super(Shekel, self).__init__(name, minimum, argmin, _box(0., 10., 4), **kwargs)
        self.defaultHP = {GaussianKernel_iso: [4.750, 1], GaussianKernel_ard: [5.146, 4.189, 4.622, 5.843, 1],
                          MaternKernel3: [15.0, 1]}
        self.A = _SHEKEL_A.copy()
        self.C = _SHEKEL_C.copy()

    def _cost(self, X):
        gap = X[:, None, :] - self.A[None, :self.wells, :]
        return -np.sum(1. / (np.sum(gap * gap, axis=2) + self.C[:self.wells]), axis=1)


class Shekel5(Shekel):
    wells = 5

    def __init__(self, **kwargs):
        super(Shekel5, self).__init__("Shekel 5", -10.1532, np.array([4.0] * 4), **kwargs)
        self.defaultHP = {GaussianKernel_iso: [0.245], GaussianKernel_ard: [0.245] * 4}


class Shekel7(Shekel):
    wells = 7

    def __init__(self, **kwargs):
        super(Shekel7, self).__init__("Shekel 7", -10.4029, np.array([4.0] * 4), **kwargs)


class Shekel10(Shekel):
    wells = 10

    def __init__(self, **kwargs):
        super(Shekel10, self).__init__("Shekel 10", -10.5364, np.array([4.0] * 4), **kwargs)
        self.defaultHP = {GaussianKernel_iso: [0.9]}


# --------------------------------------------------------------------------- 2-D classics (:210-251)
class Camelback(TestFunction):
    """six-hump camelback; global minima at (-0.0898, 0.7126) and (0.0898, -0.7126)"""

    def __init__(self, **kwargs):
        super(Camelback, self).__init__("6-Hump Camelback", -1.032, None, np.array([[-2, 2], [-1, 1]], dtype=float), **kwargs)
        self.defaultHP = {GaussianKernel_iso: [0.384], GaussianKernel_ard: [0.393, 0.387, 1], MaternKernel3: [0.842, 1]}

    def _cost(self, X):
        a, b = X[:, 0], X[:, 1]
        return (4 - 2.1 * a ** 2 + a ** 4 / 3) * a ** 2 + a * b + (-4 + 4 * b ** 2) * b ** 2


class Branin(TestFunction):
    """Branin (three global minima), divided by 100 as in the reference"""

    def __init__(self, **kwargs):
        super(Branin, self).__init__("Branin", 0.004, np.array([3.142, 4.275]), np.array([[-5, 10], [0, 15]], dtype=float),
                                     **kwargs)
        self.defaultHP = {GaussianKernel_iso: [3.8], GaussianKernel_ard: [3.4, 10.0]}

    def _cost(self, X):
        a, b = X[:, 0], X[:, 1]
        y = (b - 2 - 5.1 / (4 * np.pi ** 2) * a ** 2 + 5 / np.pi * a - 6) ** 2 + 10 * (1 - 1 / (8 * np.pi)) * np.cos(a) + 10
        return y / 100


class GoldsteinPrice(TestFunction):
    """log of the Goldstein-Price function on [-2, 2]^2 (:126-147)"""

    def __init__(self, **kwargs):
        super(GoldsteinPrice, self).__init__("Goldstein-Price", 1.0986, np.ones(2), _box(-2., 2., 2), **kwargs)
        self.defaultHP = {GaussianKernel_iso: [0.376], GaussianKernel_ard: [0.428, 0.383], MaternKernel3: [0.888, 1]}

    def _cost(self, X):
        a, b = X[:, 0], X[:, 1]
        p = 1 + (a + b + 1) ** 2 * (19 - 14 * a + 3 * a ** 2 - 14 * b + 6 * a * b + 3 * b ** 2)
        q = 30 + (2 * a - 3 * b) ** 2 * (18 - 32 * a + 12 * a ** 2 + 48 * b - 36 * a * b + 27 * b ** 2)
        return np.log(p * q)


# --------------------------------------------------------------------------- Hartman family (:254-304)
class _Hartman(TestFunction):
    """-sum_i c_i exp(-sum_j a_ij (x_j - b_ij)^2) on the unit cube"""

    def _cost(self, X):
        gap = X[:, None, :] - self.B[None, :, :]
        return -np.sum(self.C * np.exp(-np.sum(self.A * gap * gap, axis=2)), axis=1)


class Hartman3(_Hartman):
    def __init__(self, **kwargs):
        super(Hartman3, self).__init__("Hartman 3", -3.86278, np.array([.114614, .555649, 0.852547]), _box(0., 1., 3), **kwargs)
        self.defaultHP = {GaussianKernel_iso: [0.225], GaussianKernel_ard: [1.1, 0.31, 0.17]}
        self.A = np.array([[3, 10, 30], [.1, 10, 35], [3, 10, 30], [.1, 10, 35]], dtype=float)
        self.B = np.array([[0.3689, 0.1170, 0.2673], [0.4699, 0.4387, 0.7470], [0.1091, 0.8732, 0.5547],
                           [0.03825, 0.5743, 0.8828]])
        self.C = np.array([1, 1.2, 3, 3.2])


class Hartman6(_Hartman):
    def __init__(self, **kwargs):
        super(Hartman6, self).__init__("Hartman 6", -3.3224, np.array([0.2017, 0.15, 0.4769, 0.2753, 0.3117, 0.6573]),
                                       _box(0., 1., 6), **kwargs)
        self.defaultHP = {GaussianKernel_iso: [0.39], GaussianKernel_ard: [0.53, 0.57, 2.5, 0.34, 0.27, 0.35]}
        self.A = np.array([[10, 3, 17, 3.5, 1.7, 8], [0.05, 10, 17, 0.1, 8, 14], [3, 3.5, 1.7, 10, 17, 8],
                           [17, 8, 0.05, 10, 0.1, 14]], dtype=float)
        self.B = np.array([[0.1312, 0.1696, 0.5569, 0.0124, 0.8283, 0.5886], [0.2329, 0.4135, 0.8307, 0.3736, 0.1004, 0.9991],
                           [0.2348, 0.1451, 0.3522, 0.2883, 0.3047, 0.6650], [0.4047, 0.8828, 0.8732, 0.5743, 0.1091, 0.0381]])
        self.C = np.array([1, 1.2, 3, 3.2])


# --------------------------------------------------------------------------- d-dimensional bowls (:384-420)
class Sphere(TestFunction):
    def __init__(self, d=4, **kwargs):
        super(Sphere, self).__init__("Sphere %d" % d, 0, np.zeros(d), _box(-5.12, 5.12, d), **kwargs)
        self.d = d

    def _cost(self, X):
        return np.sum(X * X, axis=1)


class _IsoOnly(TestFunction):
    """d-dimensional functions whose only tuned hyper-parameter is an isotropic length scale per dimension count"""
    iso_by_d = {}

    def createKernel(self, Kernel):
        if Kernel == GaussianKernel_iso and self.d in self.iso_by_d:
            return Kernel(np.array([self.iso_by_d[self.d]]))
        raise ValueError('test function %s has no default values for kernel %s' % (self.name, Kernel.__name__))


class SumSquares(_IsoOnly):
    """sum_i i x_i^2 on [-10, 10]^d (:399-420)"""
    iso_by_d = {2: 2.42, 4: 5.3, 8: 0.5}

    def __init__(self, d=4, **kwargs):
        super(SumSquares, self).__init__("SumSquares %d" % d, 0, np.zeros(d), _box(-10., 10., d), **kwargs)
        self.d = d

    def _cost(self, X):
        return np.sum(np.arange(1, X.shape[1] + 1) * X * X, axis=1)


class Levy(_IsoOnly):
    """d-dimensional Levy function as the reference writes it (:308-334; with z = 1 + (x - 1) / 4:
    sin^2(pi z_1) + sum_{i<d} (z_i - 1)^2 (1 + 10 sin^2(pi z_i + 1)) + (z_d - 1)^2 (1 + sin^2(2 pi z_d))).
    The reference's constructor reads an undefined name and cannot run (:314); this one stores d as it meant to."""
    iso_by_d = {2: 0.9, 4: 2.8}

    def __init__(self, d=2, **kwargs):
        super(Levy, self).__init__("Levy %d" % d, 0, np.ones(d), _box(-10., 10., d), **kwargs)
        self.d = d

    def _cost(self, X):
        z = 1 + (X - 1) / 4.0
        head, last = z[:, :-1], z[:, -1]
        s = np.sin(np.pi * z[:, 0]) ** 2 + np.sum((head - 1) ** 2 * (1 + 10 * np.sin(np.pi * head + 1) ** 2), axis=1)
        return s + (last - 1) ** 2 * (1 + np.sin(2 * np.pi * last) ** 2)


class Michalewics(_IsoOnly):
    """-sum_i sin(x_i) sin(i x_i^2 / pi)^20 on [0, pi]^d, d in {2, 5, 10} (:337-362)"""
    iso_by_d = {2: 0.28, 5: 0.68, 10: 1.36}
    _known = {2: (-1.8014, np.array([2.2029, 1.5708])), 5: (-4.687658, None), 10: (-9.66015, None)}

    def __init__(self, d=2, **kwargs):
        assert d in self._known
        self.d = d
        super(Michalewics, self).__init__("Michalewics %d" % d, self._known[d][0], self._known[d][1], _box(0., np.pi, d), **kwargs)

    def _cost(self, X):
        return -np.sum(np.sin(X) * np.sin(np.arange(1, self.d + 1) * X ** 2 / np.pi) ** 20, axis=1)


class Perm(TestFunction):
    """sum_{k=1..d} (sum_j (j^k + beta) ((x_j / j)^k - 1))^2 on [-d-1, d+1]^d, minimum 0 at (1, .., d) (:365-381)"""

    def __init__(self, d=4, beta=0.5, **kwargs):
        super(Perm, self).__init__("Perm %d" % d, 0, np.arange(1, d + 1), _box(-d - 1., d + 1., d), **kwargs)
        self.beta = beta
        self.d = d

    def _cost(self, X):
        j = np.arange(1, self.d + 1, dtype=float)
        k = j[:, None]                                               # (k, j) tables, one row per power
        inner = np.sum((j[None, :] ** k + self.beta) * ((X[:, None, :] / j) ** k - 1), axis=2)
        return np.sum(inner ** 2, axis=1)


class Zakharov(TestFunction):
    """|x|^2 + a^2 + a^4 with a = sum_i i x_i / 2 on [-5, 10]^d (:425-440)"""

    def __init__(self, d=2, **kwargs):
        super(Zakharov, self).__init__("Zakharov %d" % d, 0, np.zeros(d), _box(-5., 10., d), **kwargs)
        self.d = d
        self.defaultHP = {GaussianKernel_iso: [0.5]}

    def _cost(self, X):
        a = np.sum(0.5 * np.arange(1, self.d + 1) * X, axis=1)
        return np.sum(X * X, axis=1) + a ** 2 + a ** 4


# --------------------------------------------------------------------------- a random draw from a GP (:470-527)
class Synthetic(TestFunction):
    """A function sampled from a GP prior: NX latin-hypercube sites are visited in turn, each receiving a draw
    from the running posterior (plus observation noise); the objective is that GP's posterior mean.  With
    `xstar` a minimum is planted there first and the sample sites are shifted afterwards so that the mean's
    local minimiser (BFGS from the lowest site) lands on it.  `seed` makes the construction reproducible
    (the reference draws from the global NumPy stream)."""

    def __init__(self, kernel, bounds, NX, noise=0.05, xstar=None, seed=None, device=None, **kwargs):
        super(Synthetic, self).__init__("Synthetic %d" % len(bounds), 0, None, bounds, **kwargs)
        from scipy.optimize import fmin_bfgs
        rs = np.random.RandomState(seed) if seed is not None else np.random
        sites = lhcSample(bounds, NX, seed=seed)
        gp = GaussianProcess(kernel, device=device)
        gp.addData([sites[0]], [rs.normal(0, 1)])
        floor = None
        if xstar is not None:
            floor = min(gp.Y[0] - 1.0, -2.0)
            gp.addData(np.asarray(xstar, dtype=float), floor)
        for x in sites[1:]:
            mu, s2 = gp.posterior(x)
            y = rs.normal(mu, np.sqrt(s2)) + rs.normal(0, noise)
            if floor is not None:
                y = max(y, floor + .5)               # nothing may undercut the planted minimum
            gp.addData(x, y)
        low = gp.X[int(np.argmin(gp.Y))]
        xopt = fmin_bfgs(gp.mu, low, disp=False)
        if xstar is not None:
            shifted = gp.X + (np.asarray(xstar, dtype=float) - xopt)
            gp = GaussianProcess(kernel, shifted, gp.Y.copy(), device=device)
            xopt = np.asarray(xstar, dtype=float)
        self.GP = gp
        self.minimum = gp.mu(xopt)
        self.xstar = xopt

    def values(self, X):
        y = self.GP.posteriors(np.atleast_2d(np.asarray(X, dtype=float)))[0]
        self.minimum = min(self.minimum, float(np.min(y)))
        return -y if self.maximize else y

    def f(self, x):
        return float(self.values(np.asarray(x, dtype=float).reshape(1, -1))[0])


def learnHyper(tf, Kernel, seed=None):
    """a length scale for `Kernel` on test function `tf`: BFGS on the marginal likelihood of 40 D latin-hypercube
    samples, from log 0.5 (:531-539); value and gradient come from one device factorisation per step"""
    from scipy.optimize import fmin_bfgs
    from ..gaussianprocess.trainhyper import nlml, dnlml
    D = len(tf.bounds)
    X = np.array(lhcSample(tf.bounds, D * 40, seed=seed))
    Y = tf.values(X)
    loghyper = fmin_bfgs(nlml, np.log(np.ones(1) * .5), dnlml, args=(Kernel, X, Y), disp=False)
    return np.exp(loghyper)


def checkMinimum(testfuncs, samples=100, seed=None):
    """try to undercut each function's recorded minimum: BFGS from its recorded argmin and `samples` latin-hypercube
    points (:543-555).  Prints the reference's report and returns [(name, best value found, where)]."""
    from scipy.optimize import fmin_bfgs
    found = []
    for tf in testfuncs:
        best, where = np.inf, None
        if tf.argmin is not None:
            x0 = np.asarray(tf.argmin, dtype=float)
            x1 = fmin_bfgs(tf.f, x0, disp=False)
            print('[%s] was told argmin = %s, min = %.2f' % (tf.name, tf.argmin, tf.minimum))
            print('[%s] check argmin = %s, min = %.2f' % (tf.name, tf.argmin, tf.f(x0)))
            print('[%s] found argmin = %s, min = %.2f' % (tf.name, x1, tf.f(x1)))
            best, where = tf.f(x1), x1
        P = np.array(lhcSample(tf.bounds, samples, seed=seed))
        v = tf.values(P)
        for x, y in zip(P, v):
            if y < tf.minimum:
                print('sample x = %s, y = %.4f is lower than minimum %.4f' % (x, y, tf.minimum))
        if v.min() < best:
            best, where = float(v.min()), P[int(np.argmin(v))]
        found.append((tf.name, best, where))
    return found


def plot2D(tf, N=50):
    """filled contour plot of a two-dimensional test function on an (N + 1)^2 grid (:558-577); needs matplotlib"""
    import matplotlib.pyplot as plt
    b = np.asarray(tf.bounds, dtype=float)
    c0 = np.linspace(b[0, 0], b[0, 1], N + 1)
    c1 = np.linspace(b[1, 0], b[1, 1], N + 1)
    G0, G1 = np.meshgrid(c0, c1)
    z = tf.values(np.column_stack([G0.ravel(), G1.ravel()])).reshape(G0.shape)
    fig = plt.figure(1)
    fig.clf()
    ax = fig.add_subplot(111)
    cs = ax.contourf(c0, c1, z, 50, alpha=0.9, cmap=plt.cm.jet)
    fig.colorbar(cs)
    if tf.argmin is not None:
        ax.plot(tf.argmin[0], tf.argmin[1], 'wo')
    ax.set_xbound(b[0, 0], b[0, 1])
    ax.set_ybound(b[1, 0], b[1, 1])
    return fig
