#!/usr/bin/env python3
"""
bench.py -- EI evaluations/second of the fused candidate sweep (+ GP-fit ms).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY 8d "C2"): N=1024 observations, D=4,
squared-exponential ARD kernel theta=0.3, noise 0.1, X = RandomState(2).rand, Y =
sin(3 sum X)+0.01 randn; EI (xi=.01, libm erf, native clamp) over 2^20 candidates
per GPU = RandomState(102+rank).rand, already resident in HBM when the timed region
starts.  A "step" is one full sweep of the rank's candidates, ending with the
(best value, best index) on the host -- and, for N>1, the one RCCL arg-max
exchange.  Weak scaling: every rank sweeps its own 2^20 candidates (global M =
N * 2^20), the fitted GP is replicated (each rank fits it itself).

One JSON line on rank 0; `value` = all ranks' candidates * steps / max-over-ranks time.
roofline: the sweep kernel is fp64-MFMA bound; algorithmic flops per evaluation
F = N^2 + 3ND + 4N, achieved = F * M / mean kernel time measured with HIP
events on the kernel's own stream inside the timed region.
cpu_baseline (rank 0, N=1 only): the reference's own compiled C++ (oracle/_ref/libego.so,
negei through acqmaxGP with every dimension fixed = exactly one evaluation per call) on a
bounded sample of the same candidates; falls back to the plain-C port if _ref is absent.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_OBS, DIM, M_PER_GPU = 1024, 4, 1 << 20
FP64_PEAK_TFLOPS = 78.6          # MI355X fp64 matrix = vector peak (AMD spec; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)


def synth(seed, N, D):
    rs = np.random.RandomState(seed)
    X = rs.rand(N, D)
    Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    return X, Y


def cpu_baseline(X, Y, cand, budget_s=12.0):
    """reference-shaped CPU evaluations/s on this host, 1 core (the reference is single-threaded)"""
    from oracle import oracle as orc
    ogp = orc.GP(orc.Kern("ard", [.3] * DIM), X, Y, noise=.1)
    invR = ogp.inv_factor()
    n = 0
    if orc.RefLib.available():
        ref = orc.RefLib()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s and n < len(cand):
            q = cand[n]
            ref.acqmax(ogp, [[v, v] for v in q], orc.ACQ_EI, .01, maxiter=0, invR=invR)
            n += 1
        dt = time.perf_counter() - t0
        kind = "reference"
        what = "%d candidates through oracle/_ref/libego.so acqmaxGP (lb==ub, maxiter=0: one negei call each)" % n
    else:
        chunk = 256
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s and n < len(cand):
            orc.sweep_native(ogp, cand[n:n + chunk], orc.ACQ_EI, .01, invR=invR)
            n += chunk
        dt = time.perf_counter() - t0
        kind = "port"
        what = "%d candidates through oracle/ibo_oracle.c orc_sweep_native" % n
    out = {"value": n / dt, "unit": "EI evals/s", "cores": 1, "kind": kind, "sample": what}
    # what all host cores can do with the obvious algebra (alpha cached, triangular L^-1, OpenMP over
    # candidates): the plain-C port, ~5 s of work
    m = 2048
    t0 = time.perf_counter()
    r = orc.sweep_fast(ogp, cand[:m], orc.ACQ_EI, .01)
    dt = time.perf_counter() - t0
    while dt < 4.0 and m < len(cand):
        m = min(len(cand), m * 4)
        t0 = time.perf_counter()
        r = orc.sweep_fast(ogp, cand[:m], orc.ACQ_EI, .01)
        dt = time.perf_counter() - t0
    out["best_effort_all_cores"] = {"value": m / dt, "unit": "EI evals/s", "cores": r["threads"], "kind": "port",
                                    "sample": "%d candidates through oracle/ibo_oracle.c orc_sweep_fast (OpenMP)" % m}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON: everything else that writes to file descriptor 1 (RCCL prints a
    # version / hostname / library-path banner through C stdio, flushed at exit) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    os.environ["IBO_DEVICE"] = str(local_rank)
    if world > 1:
        # one node: RCCL's bootstrap sockets go over loopback (the container hostname may not
        # resolve); the payload itself travels over xGMI peer-to-peer
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import ibo_amd
    from ibo_amd import _lib, DeviceArray
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.acquisition import sweep
    from ibo_amd.multigpu import RcclArgmax, exchange_unique_id

    # No torch in this process: torch bundles its own HIP/HSA runtime and two GPU runtimes in one
    # process do not coexist (second one finds no device / heap corruption at exit).  The launcher
    # (torch.distributed.run) only provides RANK/LOCAL_RANK/WORLD_SIZE/MASTER_PORT; rendezvous is a
    # file in /tmp keyed by the launcher's pid, barriers and the max-over-ranks are RCCL collectives.
    _lib.check(_lib.lib.ibo_device_synchronize(local_rank))
    comm = None
    id_path = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run (also with one rank)
        import threading

        def _stuck():
            sys.stderr.write("bench.py: RCCL communicator setup did not finish in 300 s (rank %d of %d)\n" % (rank, world))
            sys.stderr.flush()
            os._exit(3)
        watchdog = threading.Timer(300.0, _stuck)   # a rendezvous that never completes must not hang the node
        watchdog.daemon = True
        watchdog.start()
        uid, id_path = exchange_unique_id(world, rank)
        comm = RcclArgmax(world, rank, uid, device=local_rank)
        comm.barrier()
        watchdog.cancel()

    def barrier():
        _lib.check(_lib.lib.ibo_device_synchronize(local_rank))
        if comm is not None:
            comm.barrier()
        _lib.check(_lib.lib.ibo_device_synchronize(local_rank))

    X, Y = synth(2, N_OBS, DIM)
    GP = GaussianProcess(GaussianKernel_ard([.3] * DIM), X, Y, noise=.1, device=local_rank)
    fit_ms = []
    for _ in range(5):                                  # GP-fit ms: host X,Y -> K, L, L^-1, alpha on device
        t0 = time.perf_counter()
        GP._fit_device()
        fit_ms.append((time.perf_counter() - t0) * 1e3)
    fit_dev_ms = GP.last_fit_ms()

    cand_host = np.random.RandomState(102 + rank).rand(M_PER_GPU, DIM)
    cand = DeviceArray.from_host(cand_host, local_rank)
    start = rank * M_PER_GPU

    def step():
        r = sweep(GP, cand, acq='ei', xi=.01, native=True, index_base=start)
        if comm is not None:
            x = cand_host[r["best_idx"] - start]
            v, i, _, _ = comm.argmax(r["best_val"], r["best_idx"], x)
            return v, i, r["kernel_ms"]
        return r["best_val"], r["best_idx"], r["kernel_ms"]

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kms = []
    best = None
    for _ in range(args.steps):
        v, i, ms = step()
        kms.append(ms)
        best = (v, i)
    barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.argmax(elapsed, rank)[0]     # max over ranks (every rank gets it)

    if rank == 0:
        total = float(M_PER_GPU) * world * args.steps
        f_eval = N_OBS ** 2 + 3 * N_OBS * DIM + 4 * N_OBS
        kmean = float(np.mean(kms)) * 1e-3
        achieved = f_eval * M_PER_GPU / kmean / 1e12
        out = {
            "metric": "EI evals/sec (N=1024 obs, D=4, SE-ARD, 2^20 candidates/GPU sweep)",
            "value": total / elapsed, "unit": "EI evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: N=1024 obs, D=4, GaussianKernel_ard(0.3), noise 0.1, EI xi=0.01 (libm erf), "
                                   "2^20 candidates per GPU resident in HBM, arg-max to host each step",
                       "n_obs": N_OBS, "dim": DIM, "candidates_per_gpu": M_PER_GPU,
                       "parallelism": "candidate-sharded x%d, replicated GP, 1 RCCL arg-max exchange/step" % world},
            "gp_fit_ms": {"host_to_ready_median": float(np.median(fit_ms)), "device_events": fit_dev_ms,
                          "N": N_OBS, "D": DIM},
            "best": {"value": best[0], "index": int(best[1])},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None,
                         "kernel": "sweep_mfma_kernel", "kernel_ms": kmean * 1e3,
                         "flops_per_eval": f_eval, "evals_per_launch": M_PER_GPU},
        }
        pmc = os.path.join(ROOT, "profiles", "r01_sweep_pmc.json")
        if os.path.exists(pmc):
            # HBM bytes per launch cannot be read live; they come from the committed rocprofv3 --pmc
            # passes over this same command (tools/profile_bench.sh), FETCH_SIZE x2 (gfx950) + WRITE_SIZE
            j = json.load(open(pmc))
            out["roofline"]["traffic"] = j.get("hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = "profiles/r01_sweep_pmc.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)"
            out["roofline"]["algorithmic_bytes_per_launch"] = (8 * DIM + 16) * M_PER_GPU
            out["roofline"]["mfma_util_pct_pmc"] = j.get("mfma_util_pct")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X, Y, cand_host)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.barrier()
        comm.close()
        if rank == 0 and id_path:
            try:
                os.unlink(id_path)
            except OSError:
                pass


if __name__ == "__main__":
    main()
