#!/usr/bin/env python3
"""A/B of the two large-batch sweep kernels in one process: sweep2_kernel (sweep_variant 4) against
sweep_mfma_kernel (variant 2) -- per-candidate mu / s2 / acq, arg-max, and kernel time.
`python tools/check_sweep2.py [quick]`"""
import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ibo_amd
from ibo_amd import _lib, DeviceArray
from ibo_amd.gaussianprocess import GaussianProcess
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, GaussianKernel_iso, MaternKernel3, MaternKernel5
from ibo_amd.acquisition import sweep


def synth(seed, N, D):
    rs = np.random.RandomState(seed)
    X = rs.rand(N, D)
    return X, np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)


def variant(v):
    _lib.check(_lib.lib.ibo_set_option(b"sweep_variant", v))


def check(N, D, kern, M, tag):
    X, Y = synth(N + D, N, D)
    GP = GaussianProcess(kern, X, Y, noise=.1)
    cand = np.random.RandomState(N).rand(M, D)
    cand[min(M - 1, 77)] = X[min(N - 1, 5)]
    dc = DeviceArray.from_host(cand)
    variant(2); r0 = sweep(GP, dc, outputs=("mu", "s2", "acq"))
    variant(4); r1 = sweep(GP, dc, outputs=("mu", "s2", "acq"))
    assert r1["kernel"] == "sweep2_kernel", r1["kernel"]
    emu = np.max(np.abs(r1["mu"] - r0["mu"]) / (np.abs(r0["mu"]) + 1e-9))
    es2 = np.max(np.abs(r1["s2"] - r0["s2"]) / r0["s2"])
    eac = np.max(np.abs(r1["acq"] - r0["acq"]) / (np.abs(r0["acq"]) + 1e-12))
    ok = emu < 1e-8 and es2 < 1e-9 and eac < 1e-6 and r0["best_idx"] == r1["best_idx"]
    print("%-28s N=%5d D=%2d M=%7d  rel err mu %.1e s2 %.1e acq %.1e  argmax %d/%d  %s" %
          (tag, N, D, M, emu, es2, eac, r0["best_idx"], r1["best_idx"], "ok" if ok else "MISMATCH"), flush=True)
    return ok


def bench(N, D, kern, M, tag, reps=5):
    X, Y = synth(2, N, D)
    GP = GaussianProcess(kern, X, Y, noise=.1)
    dc = DeviceArray.from_host(np.random.RandomState(102).rand(M, D))
    F = N * N + 3 * N * D + 4 * N
    out = []
    for v in (2, 4, 2, 4):
        variant(v)
        sweep(GP, dc)
        ms = np.mean([sweep(GP, dc)["kernel_ms"] for _ in range(reps)])
        out.append("v%d %.2f ms %.1f TF (%.1f%%)" % (v, ms, F * M / ms / 1e9, F * M / ms / 1e9 / 78.6 * 100))
    print("%-20s N=%5d D=%2d M=2^%d : %s" % (tag, N, D, int(np.log2(M)), " | ".join(out)), flush=True)


if __name__ == "__main__":
    ok = True
    for N, D, kern, M, tag in [(1024, 4, GaussianKernel_ard([.3] * 4), 20000, "SE-ard"), (200, 3, GaussianKernel_iso([.4]), 9001, "SE-iso small"),
                               (64, 1, GaussianKernel_iso([.4]), 8500, "D=1"), (1000, 6, MaternKernel3([.6, 1.0]), 10000, "M3"),
                               (2048, 8, MaternKernel5([.5, 1.0]), 16384, "M5 two panels"), (1500, 5, GaussianKernel_ard([.3] * 5), 9000, "short first panel"),
                               (700, 10, GaussianKernel_ard([.5] * 10), 9000, "D=10"), (600, 13, MaternKernel5([1.0, 1.0]), 9000, "D=13"),
                               (1100, 16, GaussianKernel_ard([.9] * 16), 9000, "D=16"), (4096, 16, GaussianKernel_ard([.9] * 16), 8704, "N=4096")]:
        ok &= check(N, D, kern, M, tag)
    print("ALL OK" if ok else "FAILURES")
    if len(sys.argv) < 2:
        bench(1024, 4, GaussianKernel_ard([.3] * 4), 1 << 20, "C2")
        bench(2048, 8, MaternKernel5([.5, 1.0]), 1 << 19, "C3 shard")
        bench(1024, 16, GaussianKernel_ard([.9] * 16), 1 << 18, "D=16 SE")
        bench(1024, 12, MaternKernel5([.9, 1.0]), 1 << 18, "D=12 M5")
        bench(4096, 16, GaussianKernel_ard([.9] * 16), 1 << 17, "N=4096")
        bench(512, 4, GaussianKernel_ard([.3] * 4), 1 << 20, "N=512")
        bench(256, 4, GaussianKernel_ard([.3] * 4), 1 << 20, "N=256")
