"""
One rank of the sharded paths (ibo_amd/multigpu.py) on device 0, over the socket transport: started `world` times by
tests/rank_launcher.py, which sets RANK / WORLD_SIZE.  Every rank builds the same model and the same candidate array from
seeds, keeps its contiguous block of rows in HBM and runs

  1. sharded_sweep        (EI over its block, index_base = the block's first row, one arg-max exchange)
  2. sharded_gallery      (kept per-candidate state per shard, lock-step hallucination through the exchange)
  3. sharded_nlml_grid    (theta-points in contiguous blocks, one sum exchange, a theta that is not positive definite)

and writes what it got to <out>.rank<r>.npz.  The test compares the ranks with each other and with the single-process run.
usage: two_rank_worker.py <socket path> <out prefix> <shape: c3 | small | small8>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np


def shapes(name):
    if name == "c3":      # BASELINE config 3's model on half a GPU-shard per rank; 2 x 8 theta-points at N = 1024
        return dict(N=2048, D=8, kind="m5", hyp=[.5, 1.0], M=1 << 19, picks=8, gN=1024, gD=4, T=16, bad=11, maxiter=50)
    if name == "small8":  # eight uneven blocks: 80003 rows and 19 theta-points over 8 ranks (blocks of 10001 / 10000 rows and 3 / 2 points)
        return dict(N=600, D=3, kind="ard", hyp=[.25, .3, .35], M=80003, picks=5, gN=300, gD=3, T=19, bad=11, maxiter=12)
    return dict(N=600, D=3, kind="ard", hyp=[.25, .3, .35], M=30001, picks=5, gN=300, gD=3, T=7, bad=4, maxiter=12)


def model_and_candidates(sh):
    from conftest import synth
    from ibo_amd.gaussianprocess import GaussianProcess
    from ibo_amd.gaussianprocess import kernel as K
    X, Y = synth(900 + sh["N"], sh["N"], sh["D"])
    kern = K.MaternKernel5(sh["hyp"]) if sh["kind"] == "m5" else K.GaussianKernel_ard(sh["hyp"])
    GP = GaussianProcess(kern, X, Y, noise=.1, device=0)
    cand = np.random.RandomState(901 + sh["N"]).rand(sh["M"], sh["D"])
    return GP, cand


def grid_problem(sh):
    from conftest import synth
    X, Y = synth(77, sh["gN"], sh["gD"])
    th = np.exp(np.random.RandomState(78).uniform(np.log(.3), np.log(2), size=(sh["T"], sh["gD"])))
    th[sh["bad"]] = 3e3                                  # numerically singular with the tiny noise: NaN in that slot
    return X, Y, th, 1e-14


def main():
    address, out, shape = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["IBO_DEVICE"] = "0"                       # every rank on the one device
    from ibo_amd import DeviceArray
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.multigpu import SocketComm, shard_bounds, sharded_gallery, sharded_nlml_grid, sharded_sweep
    comm = SocketComm(world, rank, address)
    sh = shapes(shape)
    GP, cand = model_and_candidates(sh)
    a, b = shard_bounds(len(cand), world, rank)
    local = DeviceArray.from_host(cand[a:b], 0)
    r = sharded_sweep(GP, local, a, comm, acq='ei', xi=.3, native=True)
    trace = []
    gal = np.array(sharded_gallery(GP, [[0., 1.]] * sh["D"], sh["picks"], local, a, comm, maxiter=sh["maxiter"], trace=trace))
    X, Y, th, nz = grid_problem(sh)
    vals, am = sharded_nlml_grid(GaussianKernel_ard, th, X, Y, comm, noise=nz, device=0)
    comm.barrier()
    np.savez(out + ".rank%d.npz" % rank, start=a, stop=b,
             sweep=np.r_[r["best_val"], float(r["best_idx"]), float(r["best_rank"]), r["best_x"]],
             local_idx=float(r["local"]["best_idx"]), gallery=gal, nlml=vals, argmin=am,
             kernels=np.array([t["kernel"] for t in trace]), tiles=np.array([[t["tiles"], t["complete"]] for t in trace]),
             sources=np.array([t["source"] for t in trace]), sweep_idx=np.array([t["sweep_idx"] for t in trace], dtype=np.int64))
    comm.close()


if __name__ == "__main__":
    main()
