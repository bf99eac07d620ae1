"""
Mean priors for the GP (ego/gaussianprocess/prior.py).

RBFNMeanPrior.mu is part of the acqmaxGP ABI and is evaluated on the GPU inside the
sweep epilogue (csrc/sweep.hip prior_mu_dev); `train` fits the network offline:
k-means for the centres on the host (a few hundred points), the kernel matrix of the
training set and the regularised solves on the device.
"""
import numpy as np

from .. import _lib


class GPMeanPrior(object):
    def mu(self, x):
        raise NotImplementedError('GPMeanPrior-derived class does not have mean function implemented')


class RBFNMeanPrior(GPMeanPrior):
    """m(x) = sum_i beta_i exp(-theta |(x-lowerb)/width - mean_i|^2)  (prior.py:46-74)"""

    def __init__(self, means=None, beta=None, theta=10., lowerb=None, width=None):
        super(RBFNMeanPrior, self).__init__()
        self.means = means
        self.beta = beta
        self.theta = theta
        self.lowerb = lowerb
        self.width = width

    def _basis(self, U):
        """RBF activations of unit-cube points U (n, D) -> (n, k)"""
        C = np.asarray(self.means, dtype=float).reshape(len(self.means), -1)
        gap = np.linalg.norm(U[:, None, :] - C[None, :, :], axis=2)
        return self.RBF(gap)

    def mu(self, x):
        u = (np.asarray(x, dtype=float) - self.lowerb) / self.width
        return float(np.sum(self.beta * self._basis(u.reshape(1, -1))[0]))

    def negmu(self, x):
        return -self.mu(x)

    def RBF(self, r):
        return np.exp(-self.theta * r ** 2)

    def train(self, X, Y, bounds=None, k=10, delta=100, kernel=None, seed=None, device=None):
        """fit the network to data (prior.py:76-156): (1) k-means (10 Lloyd iterations from k random data,
        an emptied cluster restarts on a random datum) in the unit cube for the centres; (2) the weights as the
        generalised-least-squares solution  beta = (H^T K^-1 H + delta^-2)^-1 H^T K^-1 Y  with H the RBF
        activations and K the kernel matrix of the training points plus a ridge of 0.1 (doubled until K
        factors).  delta^-2 is added to EVERY entry of the k x k matrix, as the reference does.  The random
        stream (shuffle, then one randint per emptied cluster) follows the reference so a seed reproduces
        its centres.  Sets and returns (means, beta)."""
        rs = np.random.RandomState(seed)
        if bounds is not None:
            self.lowerb = np.array([b[0] for b in bounds], dtype=float)
            self.width = np.array([b[1] - b[0] for b in bounds], dtype=float)
        U = (np.array([np.asarray(x, dtype=float) for x in X]) - self.lowerb) / self.width
        y = np.asarray(Y, dtype=float).reshape(-1)
        n = len(U)

        order = list(range(n))
        rs.shuffle(order)
        centres = U[order[:k]]
        for _ in range(10):
            owner = np.argmin(np.linalg.norm(U[:, None, :] - centres[None, :, :], axis=2), axis=1)
            fresh = []
            for j in range(k):
                members = U[owner == j]
                fresh.append(members.mean(0) if len(members) else U[rs.randint(n)])
            centres = np.array(fresh)

        if kernel is None:
            from .kernel import GaussianKernel_iso
            kernel = GaussianKernel_iso(np.array([.2]))
        dev = _lib.default_device() if device is None else device
        K = kernel.covMatrix(U, device=dev)
        self.means = [c for c in centres]
        H = self._basis(U)
        rhs = _lib.f64(np.vstack([H.T, y[None, :]]))            # k+1 right-hand sides, one per row
        sol = np.empty_like(rhs)
        ridge = .1
        while True:
            A = _lib.f64(K + np.eye(n) * ridge)
            try:
                _lib.check(_lib.lib.ibo_spd_solve(dev, n, _lib.dp(A), len(rhs), _lib.dp(rhs), _lib.dp(sol), None))
                break
            except _lib.NotPositiveDefinite:
                ridge *= 2
                print('LinAlgError: increase regularizer to %f' % ridge)
        KiH, Kiy = sol[:k].T, sol[k]                            # K^-1 H (n, k), K^-1 y (n,)
        G = H.T.dot(KiH) + float(delta) ** -2
        try:
            beta = np.linalg.solve(G, H.T.dot(Kiy))
        except np.linalg.LinAlgError:
            beta = np.linalg.solve(G + np.eye(k), H.T.dot(Kiy))
        self.beta = np.asarray(beta).reshape(-1)
        return self.means, self.beta
