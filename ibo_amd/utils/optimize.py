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
    """one box of a finished search, as handed out in direct(..., debug=True)'s report"""
    __slots__ = ("lb", "ub", "y", "center", "d")

    def __init__(self, lb, ub, y):
        self.lb = np.asarray(lb, dtype=float)
        self.ub = np.asarray(ub, dtype=float)
        self.y = y
        self.center = 0.5 * (self.lb + self.ub)
        self.d = float(np.linalg.norm(self.lb - self.center))


def _mid(lo, hi):
    # the reference forms a midpoint from the UPPER end, hi + (lo - hi) / 2 (optimize.py:63,225,234);
    # sizes are compared with == further down, so the expression is kept to the last bit
    return hi + (lo - hi) / 2.


class _BoxStore(object):
    """The live boxes of a DIRECT search as parallel columns (lower corner, upper corner, centre value, size)
    in insertion order.  Boxes are retired by index and the columns compacted once per sweep, which leaves
    the survivors in the order one-at-a-time removal would (children always go to the end)."""

    def __init__(self):
        self.lo, self.hi, self.val, self.size = [], [], [], []

    def __len__(self):
        return len(self.val)

    def add(self, lo, hi, val):
        ctr = [_mid(a, b) for a, b in zip(lo, hi)]
        self.lo.append(lo)
        self.hi.append(hi)
        self.val.append(val)
        self.size.append(sum([(a - c) ** 2. for a, c in zip(lo, ctr)]) ** 0.5)

    def retire(self, dead):
        if not dead:
            return
        keep = [k for k in range(len(self.val)) if k not in dead]
        for name in ("lo", "hi", "val", "size"):
            col = getattr(self, name)
            setattr(self, name, [col[k] for k in keep])

    def select(self, best):
        """indices of the potentially-optimal boxes, in insertion order (optimize.py:236-275).
        A box survives unless (a) a box of exactly its size has a smaller value, (b) some larger box is at
        least as good (slope <= 0), or (c) the smallest slope up to the larger boxes is below the largest
        slope down to the smaller ones; survivors with larger boxes above them must also pass the epsilon
        test.  Every slope towards boxes of one size is extreme at that size's smallest value (subtraction
        and division by a positive number are monotone in floating point too), so the boxes are first
        grouped by size and each candidate is compared with one number per group instead of with every
        other box -- the same decisions as the all-pairs scan, from the same floating-point quotients."""
        eps = 10e-10
        sizes = sorted(set(self.size))
        rank = dict((d, g) for g, d in enumerate(sizes))
        low = [None] * len(sizes)
        for v, d in zip(self.val, self.size):
            g = rank[d]
            if low[g] is None or v < low[g]:
                low[g] = v
        chosen = []
        for k, (v, d) in enumerate(zip(self.val, self.size)):
            g = rank[d]
            if v > low[g]:
                continue
            up = None
            for h in range(g + 1, len(sizes)):
                q = (low[h] - v) / (sizes[h] - d)
                if up is None or q < up:
                    up = q
            if up is None:
                chosen.append(k)                      # nothing larger: always divided
                continue
            if up <= 0.:
                continue
            down = None
            for h in range(g):
                q = (v - low[h]) / (d - sizes[h])
                if down is None or q > down:
                    down = q
            if down is not None and up < down:
                continue
            if best == 0:
                if v <= d * up:
                    chosen.append(k)
            elif eps <= (best - v) / abs(best) + (d / abs(best)) * up:
                chosen.append(k)
        return chosen


def direct(f, bounds, args=None, debug=False, maxiter=None, maxsample=None, maxtime=None):
    """DIRECT minimisation of y = f(x, *args) over `bounds` with the behaviour of the reference's
    pure-Python optimiser (optimize.py:58-280): same sample points in the same order, same incumbent
    (strict <), same potentially-optimal sets, same termination rules.  Returns (fmin, xmin), or
    ((fmin, xmin), report) with debug=True.  At least one of maxiter / maxsample / maxtime must be set."""
    if not (maxiter or maxsample or maxtime):
        raise ValueError("No termination criterion set!")
    extra = [] if args is None else args
    t_start = time()
    ndim = len(bounds)
    span = [(b[0], b[1] - b[0]) for b in bounds]
    boxes = _BoxStore()
    inc = [None, None]                  # incumbent value and its unit-cube location
    count = [0]
    history = []

    def probe(u):
        y = f([z * w + o for z, (o, w) in zip(u, span)], *extra)
        count[0] += 1
        if inc[0] is None or y < inc[0]:
            inc[0], inc[1] = y, list(u)
        return y

    def out_of_time():
        return bool(maxtime) and time() - t_start >= maxtime

    def trisect(lo, hi, val):
        """divide one box along its longest sides (optimize.py:195-234): two probes per longest side rank
        the sides, then each side in turn is cut in three -- the outer thirds are sampled at their centres
        and stored, the middle third (parent's value) carries on to the next side"""
        longest = max([b - a for b, a in zip(hi, lo)])
        ctr = [_mid(a, b) for a, b in zip(lo, hi)]
        ranked = []
        for i in range(ndim):
            w = hi[i] - lo[i]
            if w == longest:
                p1 = list(ctr)
                p1[i] = lo[i] + w / 3.
                p2 = list(ctr)
                p2[i] = lo[i] + 2. * w / 3.
                y1 = probe(p1)
                y2 = probe(p2)
                ranked.append((min(y1, y2), i))
        ranked.sort(key=lambda t: t[0])               # stable: ties keep the dimension order
        lo, hi = list(lo), list(hi)
        for _, i in ranked:
            w = hi[i] - lo[i]
            c1 = lo[i] + w * (1 / 3)
            c2 = lo[i] + w * (2 / 3)
            left_hi = list(hi)
            left_hi[i] = c1
            boxes.add(list(lo), left_hi, probe([_mid(a, b) for a, b in zip(lo, left_hi)]))
            right_lo = list(lo)
            right_lo[i] = c2
            boxes.add(right_lo, list(hi), probe([_mid(a, b) for a, b in zip(right_lo, hi)]))
            lo[i], hi[i] = c1, c2
        boxes.add(lo, hi, val)

    def finish():
        upper = np.array([b[1] for b in bounds], dtype=float)
        lower = np.array([b[0] for b in bounds], dtype=float)

        def to_box(u):
            return np.array(u) * (upper - lower) + lower
        best = (inc[0], to_box(inc[1]))
        if not debug:
            return best
        return best, {'fmin evolution': [(y, to_box(u)) for y, u in history],
                      'rectangles': [Rectangle(to_box(a), to_box(b), v) for a, b, v in zip(boxes.lo, boxes.hi, boxes.val)],
                      'samples': count[0]}

    trisect([0.] * ndim, [1.] * ndim, probe([.5] * ndim))
    sweeps = 0
    while True:
        sweeps += 1
        if maxiter and sweeps > maxiter:
            return finish()
        todo = boxes.select(inc[0])
        if out_of_time():
            return finish()
        # the chosen boxes leave the store as they are divided; their children join at the end
        done = set()
        stop = False
        for k in todo:
            done.add(k)
            trisect(boxes.lo[k], boxes.hi[k], boxes.val[k])
            if (maxsample and count[0] >= maxsample) or out_of_time():
                stop = True
                break
        boxes.retire(done)
        history.append((inc[0], list(inc[1])))
        if stop:
            return finish()


def cdirect(f, bounds, args=None, maxiter=10, maxtime=10, maxsample=200000, compat=True, return_samples=False,
            batched=False, **kwargs):
    """native DIRECT on a Python objective (optimize.py:310-343) -> (fmin, xmin).
    compat=True keeps the reference's dimension-0 quirk.  batched=True runs the schedule the GPU objective
    is evaluated under (one batch per iteration: probes plus verified guesses of the child centres); the result, the
    point and the sample count are those of the sequential call order, only the order of the calls to f differs."""
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
                                        int(maxsample), (1 if compat else 0) | (2 if batched else 0), ctypes.byref(fmin), _lib.dp(xmin),
                                        ctypes.byref(ns)))
    if return_samples:
        return fmin.value, xmin, ns.value
    return fmin.value, xmin
