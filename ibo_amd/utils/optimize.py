"""
DIRECT optimisers with the reference's call signatures (ego/utils/optimize.py):

    direct(f, bounds, args=None, debug=False, maxiter=None, maxsample=None, maxtime=None)
        pure-Python DIRECT on a Python objective (minimises) -> (fmin, xmin)
    cdirect(f, bounds, args=None, maxiter=10, maxtime=10, maxsample=200000)
        the native DIRECT of libibo_hip (csrc/direct_host.cpp) on a Python callback

Both are host logic (the objective is an arbitrary Python callable); the GP
acquisitions never go through here on the default path -- maximizeEI/PI/UCB run
DIRECT with batched GPU evaluation (ibo_direct_max).
"""
import ctypes
from time import time

import numpy as np

from .. import _lib


class Rectangle(object):
    def __init__(self, lb, ub, y):
        self.lb = list(lb)
        self.ub = list(ub)
        self.y = y
        # the reference zips (lb, ub) into (u, l) (optimize.py:63): centre = ub + (lb-ub)/2
        self.center = [l + (u - l) / 2. for u, l in zip(self.lb, self.ub)]
        self.d = sum([(l - c) ** 2. for l, c in zip(self.lb, self.center)]) ** 0.5


def direct(f, bounds, args=None, debug=False, maxiter=None, maxsample=None, maxtime=None):
    """DIRECT minimisation of y = f(x, *args) over `bounds` (optimize.py:68-280).
    At least one of maxiter / maxsample / maxtime must be set."""
    if not (maxiter or maxsample or maxtime):
        raise ValueError("No termination criterion set!")
    if args is None:
        args = []
    tic = time()
    state = {'fmin': None, 'samples': 0}
    fminevol = []
    N = len(bounds)
    rectangles = []          # the reference uses a set; a list only fixes the tie order

    def samplef(x):
        xprime = [z * (b[1] - b[0]) + b[0] for z, b in zip(x, bounds)]
        y = f(xprime, *args)
        state['samples'] += 1
        if state['fmin'] is None or y < state['fmin'][0]:
            state['fmin'] = [y, list(x)]
        return y

    def divrec(rect):
        rectangles.remove(rect)
        maxlength = max([u - l for u, l in zip(rect.ub, rect.lb)])
        I = []
        for i in range(N):
            if rect.ub[i] - rect.lb[i] == maxlength:
                s1 = list(rect.center)
                s2 = list(rect.center)
                iwidth = rect.ub[i] - rect.lb[i]
                s1[i] = rect.lb[i] + iwidth / 3.
                s2[i] = rect.lb[i] + 2. * iwidth / 3.
                I.append((i, min(samplef(s1), samplef(s2))))
        I.sort(key=lambda t: t[1])
        oldrect = rect
        target = rect
        for i, _ in I:
            dwidth = oldrect.ub[i] - oldrect.lb[i]
            split1 = oldrect.lb[i] + dwidth * (1 / 3)
            split2 = oldrect.lb[i] + dwidth * (2 / 3)
            lb1 = list(oldrect.lb); ub1 = list(oldrect.ub); ub1[i] = split1
            rectangles.append(Rectangle(lb1, ub1, samplef([l + (u - l) / 2. for u, l in zip(lb1, ub1)])))
            lb2 = list(oldrect.lb); ub2 = list(oldrect.ub); lb2[i] = split1; ub2[i] = split2
            target = Rectangle(lb2, ub2, oldrect.y)
            lb3 = list(oldrect.lb); ub3 = list(oldrect.ub); lb3[i] = split2
            rectangles.append(Rectangle(lb3, ub3, samplef([l + (u - l) / 2. for u, l in zip(lb3, ub3)])))
            oldrect = target
        rectangles.append(target)

    def results():
        lbounds = np.array([x[0] for x in bounds], dtype=float)
        ubounds = np.array([x[1] for x in bounds], dtype=float)

        def trans(x):
            return np.array(x) * (ubounds - lbounds) + lbounds
        optimum = (state['fmin'][0], trans(state['fmin'][1]))
        if debug:
            report = {'fmin evolution': [(y, trans(x)) for y, x in fminevol],
                      'rectangles': [Rectangle(trans(r.lb), trans(r.ub), r.y) for r in rectangles],
                      'samples': state['samples']}
            return optimum, report
        return optimum

    first = Rectangle([0.] * N, [1.] * N, samplef([.5] * N))
    rectangles.append(first)
    divrec(first)

    iteration = 0
    epsilon = 10e-10
    while True:
        iteration += 1
        if maxiter and iteration > maxiter:
            return results()
        potopts = []
        for Rj in list(rectangles):
            maxI1 = None
            minI2 = None
            broke = False
            for Ri in rectangles:
                if Ri is Rj:
                    continue
                if Ri.d < Rj.d:
                    val = (Rj.y - Ri.y) / (Rj.d - Ri.d)
                    if maxI1 is None or val > maxI1:
                        maxI1 = val
                elif Ri.d > Rj.d:
                    val = (Ri.y - Rj.y) / (Ri.d - Rj.d)
                    if minI2 is None or val < minI2:
                        minI2 = val
                        if minI2 <= 0.:
                            broke = True
                            break
                else:
                    if Rj.y > Ri.y:
                        broke = True
                        break
                if maxI1 is not None and minI2 is not None and minI2 < maxI1:
                    broke = True
                    break
            if not broke:
                F = state['fmin'][0]
                if not minI2:
                    potopts.append(Rj)
                elif F == 0:
                    if Rj.y <= Rj.d * minI2:
                        potopts.append(Rj)
                elif epsilon <= (F - Rj.y) / abs(F) + (Rj.d / abs(F)) * minI2:
                    potopts.append(Rj)
            if maxtime and time() - tic >= maxtime:
                return results()
        for Rj in potopts:
            divrec(Rj)
            if maxsample and state['samples'] >= maxsample:
                fminevol.append(state['fmin'])
                return results()
            if maxtime and time() - tic >= maxtime:
                fminevol.append(state['fmin'])
                return results()
        fminevol.append(state['fmin'])


def cdirect(f, bounds, args=None, maxiter=10, maxtime=10, maxsample=200000, compat=True, return_samples=False,
            **kwargs):
    """native DIRECT on a Python objective (optimize.py:310-343) -> (fmin, xmin).
    compat=True keeps the reference's dimension-0 quirk."""
    if args is None:
        args = []
    n = len(bounds)

    def objective(k, x):
        return float(f(np.array([x[i] for i in range(k)]), *args))

    lower = _lib.f64([b[0] for b in bounds])
    upper = _lib.f64([b[1] for b in bounds])
    fmin = ctypes.c_double()
    xmin = np.empty(n)
    ns = ctypes.c_int64()
    cb = _lib.OBJECTIVE(objective)
    _lib.check(_lib.lib.ibo_direct_host(cb, n, _lib.dp(lower), _lib.dp(upper), int(maxiter), int(maxtime),
                                        int(maxsample), 1 if compat else 0, ctypes.byref(fmin), _lib.dp(xmin),
                                        ctypes.byref(ns)))
    if return_samples:
        return fmin.value, xmin, ns.value
    return fmin.value, xmin
