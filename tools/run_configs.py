#!/usr/bin/env python3
"""Run BASELINE.json configs C2..C5 at full size on one GPU and print timings +
size-independent checks.  `python tools/run_configs.py [c2 c3 c4 c5]`"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ibo_amd
from ibo_amd import DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess, PrefGaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, GaussianKernel_iso, MaternKernel5
from ibo_amd.gaussianprocess.trainhyper import nlml_grid
from ibo_amd.acquisition import sweep, maximizeEI
from ibo_amd.acquisition.gallery import fastUCBGallery


def synth(seed, N, D):
    rs = np.random.RandomState(seed)
    X = rs.rand(N, D)
    return X, np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)


def hartman6(x):
    A = np.array([[10, 3, 17, 3.5, 1.7, 8], [0.05, 10, 17, 0.1, 8, 14], [3, 3.5, 1.7, 10, 17, 8], [17, 8, 0.05, 10, 0.1, 14]])
    P = np.array([[0.1312, 0.1696, 0.5569, 0.0124, 0.8283, 0.5886], [0.2329, 0.4135, 0.8307, 0.3736, 0.1004, 0.9991],
                  [0.2348, 0.1451, 0.3522, 0.2883, 0.3047, 0.6650], [0.4047, 0.8828, 0.8732, 0.5743, 0.1091, 0.0381]])
    C = np.array([1, 1.2, 3, 3.2])
    return float(np.sum(C * np.exp(-np.sum(A * (x - P) ** 2, axis=1))))      # maximise


def tm(f, n=1):
    t0 = time.perf_counter(); r = None
    for _ in range(n):
        r = f()
    return r, (time.perf_counter() - t0) / n * 1e3


def c2():
    X, Y = synth(2, 1024, 4)
    GaussianProcess(GaussianKernel_ard([.3] * 4), X, Y, noise=.1)              # (warm: the first model of a size allocates its buffers)
    GP, fit = tm(lambda: GaussianProcess(GaussianKernel_ard([.3] * 4), X, Y, noise=.1))
    cand = DeviceArray.from_host(np.random.RandomState(102).rand(1 << 20, 4))
    sweep(GP, cand)
    r, ms = tm(lambda: sweep(GP, cand), 5)
    _, dms = tm(lambda: maximizeEI(GP, [[0., 1.]] * 4), 3)
    ch = np.random.RandomState(102).rand(1 << 20, 4)
    GP._posterior_arrays(ch)                              # first call pins the staging buffers (one-off, ~40 ms)
    _, pms = tm(lambda: GP._posterior_arrays(ch), 3)      # host in / host out: PCIe-inclusive, steady state
    return dict(fit_ms=fit, fit_dev_ms=GP.last_fit_ms(), sweep_ms=ms, kernel_ms=r["kernel_ms"], evals_per_s=(1 << 20) / ms * 1e3,
                maximizeEI_default_ms=dms, best=r["best_idx"], posteriors_host_1M_ms=pms,
                posteriors_host_evals_per_s=(1 << 20) / pms * 1e3)


def c3(M=1 << 19):
    """N=2048, D=8, Matern-5/2 [.5, 1], gallery of 8 over one GPU's shard (4M/8 = 524288 candidates)"""
    X, Y = synth(3, 2048, 8)
    GaussianProcess(MaternKernel5([.5, 1.0]), X, Y, noise=.1)
    GP, fit = tm(lambda: GaussianProcess(MaternKernel5([.5, 1.0]), X, Y, noise=.1))
    cand = DeviceArray.from_host(np.random.RandomState(103).rand(M, 8))
    sweep(GP, cand)
    r, ms = tm(lambda: sweep(GP, cand), 3)
    _, gcold = tm(lambda: fastUCBGallery(GP, [[0., 1.]] * 8, 8, candidates=cand))
    gal, gms = tm(lambda: fastUCBGallery(GP, [[0., 1.]] * 8, 8, candidates=cand))
    gal = np.array(gal)
    dmin = min(np.linalg.norm(gal[i] - gal[j]) for i in range(8) for j in range(i))
    assert np.all(gal >= 0) and np.all(gal <= 1)
    F = 2048 ** 2 + 3 * 2048 * 8 + 4 * 2048
    return dict(fit_ms=fit, fit_dev_ms=GP.last_fit_ms(), sweep_ms=ms, kernel_ms=r["kernel_ms"], evals_per_s=M / ms * 1e3,
                tflops=F * M / r["kernel_ms"] / 1e9, gallery8_ms=gcold, gallery8_warm_ms=gms, gallery_min_dist=dmin)


def c4(P=512, M=1 << 20):
    rs = np.random.RandomState(4)
    pts = rs.rand(2 * P, 6)
    prefs = []
    for i in range(P):
        a, b = pts[2 * i], pts[2 * i + 1]
        prefs.append((a, b, 0) if hartman6(a) > hartman6(b) else (b, a, 0))
    _, fit_cold = tm(lambda: PrefGaussianProcess(GaussianKernel_ard([0.53, 0.57, 2.5, 0.34, 0.27, 0.35]), prefs))
    GP, fit = tm(lambda: PrefGaussianProcess(GaussianKernel_ard([0.53, 0.57, 2.5, 0.34, 0.27, 0.35]), prefs))
    ok = sum(GP.mu(v) > GP.mu(u) for v, u, _ in prefs[:64])
    cand = DeviceArray.from_host(np.random.RandomState(104).rand(M, 6))
    _, gcold = tm(lambda: fastUCBGallery(GP, [[0., 1.]] * 6, 8, candidates=cand))
    gal, gms = tm(lambda: fastUCBGallery(GP, [[0., 1.]] * 6, 8, candidates=cand))
    return dict(addPreferences_ms=fit, addPreferences_first_call_ms=fit_cold, gallery8_ms=gcold, gallery8_warm_ms=gms, n_points=len(GP.X), orderings_respected_of_64=int(ok))


def c5(N=4096, T=64):
    X, Y = synth(5, N, 16)
    thetas = np.exp(np.random.RandomState(105).uniform(np.log(.1), np.log(3), size=(512, 16)))[:T]
    # warm: the first call of this size allocates the batch workspace (17 GB of matrices and packed factors) and pads it
    (_, _), cold = tm(lambda: nlml_grid(GaussianKernel_ard, thetas, X, Y, noise=1e-3))
    (vals, am), ms = tm(lambda: nlml_grid(GaussianKernel_ard, thetas, X, Y, noise=1e-3), 3)
    return dict(n_theta=T, total_ms=ms, ms_per_theta=ms / T, first_call_ms=cold, n_not_pd=int(np.sum(~np.isfinite(vals))), argmin=am,
                best=float(np.nanmin(vals)))


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c3", "c4", "c5"]
    for w in which:
        t0 = time.perf_counter()
        out = globals()[w]()
        out["wall_s"] = time.perf_counter() - t0
        print(w, json.dumps(out), flush=True)
