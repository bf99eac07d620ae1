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

from ..gaussianprocess import GaussianProcess
from ..utils.latinhypercube import lhcSample
from . import EI, maximizeEI, sweep
from .. import _lib


MIN_SEPARATION = .5          # gallery members must be farther apart than this (gallery.py:102,113,125)


def _separated(x, members):
    """True when x is farther than MIN_SEPARATION from every member (vacuously for an empty gallery)"""
    if len(members) == 0:
        return True
    return float(np.min(np.linalg.norm(np.asarray(members, dtype=float) - np.asarray(x, dtype=float), axis=1))) > MIN_SEPARATION


def _best_observation_inside(X, Y, bounds):
    """the best observation seen so far that lies inside the box (first one on ties), or None"""
    box = np.asarray(bounds, dtype=float)
    X = np.asarray(X, dtype=float)
    inside = np.all((X >= box[:, 0]) & (X <= box[:, 1]), axis=1)
    score = np.where(inside & ~np.isnan(Y), Y, -np.inf)
    k = int(np.argmax(score))
    return X[k] if score[k] > -np.inf else None


def _start(GP, bounds, useBest, rounds=0):
    """first gallery member(s) and the plain GP the rounds hallucinate on (gallery.py:49-90): the best
    in-box observation when there are data; the box centre for an empty model without a prior; otherwise
    the highest posterior-mean point reached by a local search from each RBF centre of the prior"""
    # default noise: the source model may be a preference GP; room for the rounds' hallucinated rows, so that every one
    # of them is an in-place extension of the factor (and of the kept sweep state) rather than a refit
    plain = dict(prior=GP.prior, device=GP._device, reserve_rows=rounds)
    if len(GP.X) > 0:
        first = _best_observation_inside(GP.X, GP.Y, bounds) if useBest else None
        model = GaussianProcess(deepcopy(GP.kernel), np.array(GP.X, dtype=float), np.array(GP.Y, dtype=float), **plain)
        return ([] if first is None else [first]), model
    if GP.prior is None:
        centre = np.array([(lo + hi) / 2. for lo, hi in bounds])
        return [centre], GaussianProcess(deepcopy(GP.kernel), [centre], [0.0], **plain)
    from scipy.optimize import fmin_bfgs
    box = np.asarray(bounds, dtype=float)
    ends = [np.clip(fmin_bfgs(GP.negmu, m, disp=False), box[:, 0], box[:, 1]) for m in GP.prior.means]
    heights = [GP.mu(e) for e in ends]
    k = int(np.argmax(heights))                           # first of equals, as a strict > scan would pick
    return [ends[k]], GaussianProcess(deepcopy(GP.kernel), ends[k], heights[k], **plain)


def _prior_centres(model, bounds, members, floor):
    """the prior's RBF centres as extra candidates (gallery.py:118-130): clipped to the box, mapped through
    the prior's affine normalisation, scored with EI(xi=.4) in ONE batched device call; returns
    (value, point) of the best admissible one that beats `floor`, else None"""
    pr = model.prior
    box = np.asarray(bounds, dtype=float)
    pts = np.clip(np.asarray(pr.means, dtype=float).reshape(len(pr.means), -1), box[:, 0], box[:, 1])
    pts = pts * np.asarray(pr.width, dtype=float) + np.asarray(pr.lowerb, dtype=float)
    score = EI(model, xi=.4).values(pts)
    ok = np.array([_separated(x, members) for x in pts])
    score = np.where(ok, score, -np.inf)
    k = int(np.argmax(score))
    return (float(score[k]), pts[k]) if score[k] > floor else None


def _state_info(model):
    """(tiles, complete) of the sweep state kept on the model's handle (ibo_sweep_state_info)"""
    import ctypes
    t, c = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(_lib.lib.ibo_sweep_state_info(model._handle(), ctypes.byref(t), ctypes.byref(c)))
    return t.value, c.value


def fastUCBGallery(GP, bounds, N, useBest=True, samples=300, useCDIRECT=True, candidates=None, seed=None,
                   lhc_per_round=None, comm=None, index_base=0, maxiter=50, maxsample=10000, trace=None):
    """N points to show a user (gallery.py:42-136).  Each round proposes (1) the DIRECT maximiser of
    EI(xi=.3), (2) the best of a sample set under EI(xi=.4, NR erf) -- one fused sweep, distance rule applied
    in the kernel -- and (3) the prior's centres; the highest admissible proposal joins the gallery and is
    added to the model with its own posterior mean as a hallucinated observation.

    maxiter / maxsample   the DIRECT step's budgets (the reference's maximizeEI defaults)
    trace                 a list that receives one dict per round: the DIRECT proposal (opt, optx), the sweep's
                          (value, global index), which proposal won, the sweep kernel and the kept state's (tiles, complete)"""
    gallery, model = _start(GP, bounds, useBest, rounds=N)
    # a fixed candidate array is swept every round while the model grows by one hallucinated point: it goes to
    # HBM once, and from the second round on the device folds the model's new row into the per-candidate state it
    # kept (sweep(incremental=True)) instead of repeating the O(N^2)-per-candidate sweep
    fixed = candidates is not None and lhc_per_round is None
    if fixed and not isinstance(candidates, _lib.DeviceArray):
        candidates = _lib.DeviceArray.from_host(np.atleast_2d(np.asarray(candidates, dtype=float)), model._dev.device)
    rnd = 0
    while len(gallery) < N:
        pick_val, pick = -np.inf, None
        opt, optx = maximizeEI(model, bounds, xi=.3, useCDIRECT=useCDIRECT, maxiter=maxiter, maxsample=maxsample)
        source = None
        if _separated(optx, gallery):
            pick_val, pick, source = opt, optx, "direct"

        if lhc_per_round is not None:
            S = np.asarray(lhc_per_round[rnd], dtype=float)
        elif candidates is not None:
            S = candidates
        else:
            S = np.array(lhcSample(bounds, samples, seed=None if seed is None else seed + rnd))
        shown = np.array(gallery) if gallery else None
        if comm is not None:
            # S is this rank's block of the candidate array (rows index_base ...): sharded sweep + one exchange
            from ..multigpu import sharded_sweep
            r = sharded_sweep(model, S, index_base, comm, acq='ei', xi=.4, native=False, exclude=shown,
                              exclude_radius=MIN_SEPARATION, incremental=fixed)
            if r["best_idx"] >= 0 and r["best_val"] > pick_val:
                pick_val, pick, source = r["best_val"], np.array(r["best_x"]), "sweep"
        else:
            r = sweep(model, S, acq='ei', xi=.4, native=False, exclude=shown, exclude_radius=MIN_SEPARATION,
                      incremental=fixed)
            if r["best_idx"] >= 0 and r["best_val"] > pick_val:
                k = r["best_idx"]
                pick_val, source = r["best_val"], "sweep"
                pick = S.view_rows(k, k + 1).to_host()[0] if isinstance(S, _lib.DeviceArray) else np.array(S[k])

        if model.prior is not None:
            extra = _prior_centres(model, bounds, gallery, pick_val)
            if extra is not None:
                (pick_val, pick), source = extra, "prior"

        if trace is not None:
            local = r.get("local", r)
            tiles, complete = _state_info(model) if fixed else (0, 0)
            trace.append(dict(round=rnd, opt=opt, optx=np.array(optx), sweep_val=r["best_val"], sweep_idx=r["best_idx"],
                              source=source, value=pick_val, kernel=local["kernel"], tiles=tiles, complete=complete,
                              n_model=len(model.X)))
        gallery.append(pick)
        model.addData(pick, model.mu(pick))
        rnd += 1
    return gallery
