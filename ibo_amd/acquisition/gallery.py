"""
fastUCBGallery (ego/acquisition/gallery.py:42-136): greedy selection of N points to
show a user, by repeatedly maximising EI on a GP that is updated with
hallucinated observations y = mu(x).

Per round, as the reference: (1) DIRECT maximisation of EI(xi=.3) -- kept only if
farther than 0.5 from every gallery member; (2) EI(xi=.4, NR erf) over a sample
set, same distance rule; (3) the prior means as extra candidates; then
hallucGP.addData(best, mu(best)).  Step (2) is one fused GPU sweep with the
distance mask applied inside the kernel; `candidates=` swaps the 300-sample
latin hypercube for any (M, D) array (ndarray or DeviceArray already in HBM)
and `seed=` makes the default sampling reproducible (the reference is unseeded).
"""
from copy import deepcopy

import numpy as np
from numpy.linalg import norm

from ..gaussianprocess import GaussianProcess
from ..utils.latinhypercube import lhcSample
from . import EI, maximizeEI, sweep
from .. import _lib


def fastUCBGallery(GP, bounds, N, useBest=True, samples=300, useCDIRECT=True, candidates=None, seed=None,
                   lhc_per_round=None, comm=None, index_base=0):
    gallery = []
    if len(GP.X) > 0:
        if useBest:
            bestY = -np.inf
            bestX = None
            for x, y in zip(GP.X, GP.Y):
                if y > bestY:
                    for v, b in zip(x, bounds):
                        if v < b[0] or v > b[1]:
                            break
                    else:
                        bestY = y
                        bestX = x
            if bestX is not None:
                gallery.append(bestX)
        # a plain GP on the same data (the original may be a preference GP), default noise
        hallucGP = GaussianProcess(deepcopy(GP.kernel), deepcopy(GP.X), deepcopy(GP.Y), prior=GP.prior,
                                   device=GP._device)
    elif GP.prior is None:
        x = np.array([(b[0] + b[1]) / 2. for b in bounds])
        gallery.append(x)
        hallucGP = GaussianProcess(deepcopy(GP.kernel), [x], [0.0], prior=GP.prior, device=GP._device)
    else:
        from scipy.optimize import fmin_bfgs
        bestmu = -np.inf
        bestX = None
        for m in GP.prior.means:
            argmin = fmin_bfgs(GP.negmu, m, disp=False)
            for i in range(len(argmin)):
                argmin[i] = np.clip(argmin[i], bounds[i][0], bounds[i][1])
            if GP.mu(argmin) > bestmu:
                bestX = argmin
                bestmu = GP.mu(argmin)
        gallery.append(bestX)
        hallucGP = GaussianProcess(deepcopy(GP.kernel), bestX, bestmu, prior=GP.prior, device=GP._device)

    rnd = 0
    while len(gallery) < N:
        bestUCB = -np.inf
        bestX = None
        opt, optx = maximizeEI(hallucGP, bounds, xi=.3, useCDIRECT=useCDIRECT)
        if len(gallery) == 0 or min(norm(optx - gx) for gx in gallery) > .5:
            bestUCB = opt
            bestX = optx

        # sample set for this round
        if lhc_per_round is not None:
            S = np.asarray(lhc_per_round[rnd], dtype=float)
        elif candidates is not None:
            S = candidates
        else:
            S = np.array(lhcSample(bounds, samples, seed=None if seed is None else seed + rnd))
        excl = np.array(gallery) if gallery else None
        if comm is not None:
            # S is this rank's block of the candidate array (rows index_base ...): sharded sweep + one exchange
            from ..multigpu import sharded_sweep
            r = sharded_sweep(hallucGP, S, index_base, comm, acq='ei', xi=.4, native=False, exclude=excl,
                              exclude_radius=.5)
            if r["best_idx"] >= 0 and r["best_val"] > bestUCB:
                bestUCB = r["best_val"]
                bestX = np.array(r["best_x"])
        else:
            r = sweep(hallucGP, S, acq='ei', xi=.4, native=False, exclude=excl, exclude_radius=.5)
            if r["best_idx"] >= 0 and r["best_val"] > bestUCB:
                bestUCB = r["best_val"]
                if isinstance(S, _lib.DeviceArray):
                    bestX = S.view_rows(r["best_idx"], r["best_idx"] + 1).to_host()[0]
                else:
                    bestX = np.array(S[r["best_idx"]])

        if hallucGP.prior is not None:
            ut = EI(hallucGP, xi=.4)
            for x in hallucGP.prior.means:
                x = np.array([np.clip(x[i], bounds[i][0], bounds[i][1]) for i in range(len(x))])
                x = x * hallucGP.prior.width + hallucGP.prior.lowerb
                u = -ut.negf(x)
                if u > bestUCB:
                    if len(gallery) == 0 or min(norm(x - gx) for gx in gallery) > .5:
                        bestUCB = u
                        bestX = x

        gallery.append(bestX)
        hallucGP.addData(bestX, hallucGP.mu(bestX))
        rnd += 1
    return gallery
