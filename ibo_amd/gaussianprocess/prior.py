"""
Mean priors for the GP (ego/gaussianprocess/prior.py).  Only evaluation is on the
hot path -- RBFNMeanPrior.mu is part of the acqmaxGP ABI and is evaluated on the
GPU inside the sweep epilogue; the offline k-means/ridge `train` is out of scope
 and not provided.
"""
import numpy as np


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

    def mu(self, x):
        x = (np.asarray(x, dtype=float) - self.lowerb) / self.width
        norms = [np.linalg.norm(m - x) for m in self.means]
        rbf = np.array([self.RBF(n) for n in norms])
        return float(np.sum(self.beta * rbf))

    def negmu(self, x):
        return -self.mu(x)

    def RBF(self, r):
        return np.exp(-self.theta * r ** 2)

    def train(self, *args, **kwargs):
        raise NotImplementedError("RBFNMeanPrior.train (offline k-means + ridge fit) is outside the accelerated "
                                  "path; construct the prior from means/beta/theta/lowerb/width")
