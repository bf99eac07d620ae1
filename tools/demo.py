#!/usr/bin/env python3
"""
The data path of the reference's demo.py (demoObservations :59-98 and demoPrefGallery :138-195)
without the matplotlib / raw_input parts, run through the `ego.*` import paths that
ibo_amd.install_as_ego() provides.  BASELINE config 1 ("demo plumbing"): on this backend every
numerical step below executes on the GPU.
    python tools/demo.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ibo_amd                                         # noqa: E402
ibo_amd.install_as_ego()
from ego.gaussianprocess import GaussianProcess, PrefGaussianProcess          # noqa: E402
from ego.gaussianprocess.kernel import GaussianKernel_ard                      # noqa: E402
from ego.acquisition import maximizeEI, EI                                     # noqa: E402
from ego.acquisition.gallery import fastUCBGallery                             # noqa: E402


def demo_observations():
    kernel = GaussianKernel_ard(np.array([.5, .5, .3]))
    GP = GaussianProcess(kernel, noise=0.1)
    X = [np.array([1, 1.5, 0.9]), np.array([.8, -.2, -0.1]), np.array([2, .8, -.2]), np.array([0, 0, .5])]
    Y = [1, .7, .6, -.1]
    GP.addData(X, Y)
    for tx in (np.array([1, 1.45, 1.0]), np.array([-10, .5, -10])):
        mu, sig2 = GP.posterior(tx)
        print('the posterior of %s is a normal distribution N(%.3f, %.3f)' % (tx, mu, sig2))
    bound = [[1, 1], [-1.99, 1.98], [-1.99, 1.98]]       # first dimension fixed at 1
    for step in range(3):
        opt, optx = maximizeEI(GP, bound, xi=.1)
        print('step %d: max EI %.6f at %s (EI class there: %.6f)' % (step, opt, optx, EI(GP, xi=.1).f(optx)))
        GP.addData(optx, float(np.sin(optx.sum())))      # stand-in for the user's rating


def demo_pref_gallery():
    kernel = GaussianKernel_ard(np.array([.5, .5]))
    GP = PrefGaussianProcess(kernel)
    bounds = [[0., 1.], [0., 1.]]
    f = lambda x: -np.sum((np.asarray(x) - .6) ** 2)     # hidden utility standing in for the user
    gallery = fastUCBGallery(GP, bounds, 2, seed=1)
    for rnd in range(3):
        best = max(gallery, key=f)
        prefs = [(best, g, 0) for g in gallery if g is not best]
        GP.addPreferences(prefs)
        gallery = fastUCBGallery(GP, bounds, 4, seed=10 + rnd)
        print('round %d: %d preferences, gallery %s' % (rnd, len(GP.preferences), np.round(np.array(gallery), 3).tolist()))


if __name__ == "__main__":
    demo_observations()
    demo_pref_gallery()
