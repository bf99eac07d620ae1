"""
The sharded paths with MORE THAN ONE RANK on the one GPU of the box (ego/acquisition/gallery.py:93-134 is the
loop they shard).  RCCL refuses two ranks on one device, so the ranks talk through ibo_amd.multigpu.SocketComm -- the same
slot protocol, the same final reduction -- and everything else is the product path: per-rank handles, candidate blocks with
index_base != 0, the kept sweep state per shard, lock-step hallucination, the theta blocks of the NLML grid.

The rank processes are fresh interpreters started by tests/rank_launcher.py (a helper that never touches the GPU; this
pytest process has).  Every rank must return the same result, and that result must equal the single-process run BIT FOR
BIT.  A failing rank exits non-zero and the launcher terminates the others.
"""
import os
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("world,shape", [(2, "c3"), (3, "small"), (8, "small8")])
def test_sharded_paths_with_several_ranks_on_one_gpu(launch_ranks, world, shape):
    import two_rank_worker as W
    from ibo_amd import DeviceArray, _lib
    from ibo_amd.acquisition import sweep
    from ibo_amd.acquisition.gallery import fastUCBGallery
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
    from ibo_amd.gaussianprocess.trainhyper import nlml_grid
    from ibo_amd.multigpu import shard_bounds
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the product has no CPU fallback")
    with tempfile.TemporaryDirectory(prefix="ibo_ranks_") as tmp:
        out = os.path.join(tmp, "res")
        res = launch_ranks([os.path.join(ROOT, "tests", "two_rank_worker.py"), os.path.join(tmp, "sock"), out, shape], world,
                           timeout=900)
        assert res["rc"] == [0] * world, "\n".join("rank %d rc %s\n%s" % (r, rc, e[-3000:]) for r, (rc, e) in enumerate(zip(res["rc"], res["err"])))
        ranks = [dict(np.load(out + ".rank%d.npz" % r)) for r in range(world)]
    sh = W.shapes(shape)
    # the blocks tile the array, rank 1's block does not start at row 0
    for r in range(world):
        assert (int(ranks[r]["start"]), int(ranks[r]["stop"])) == shard_bounds(sh["M"], world, r)
    assert int(ranks[1]["start"]) > 0
    # every rank returns the same result ...
    for r in range(1, world):
        for key in ("sweep", "gallery", "nlml", "argmin"):
            np.testing.assert_array_equal(ranks[r][key], ranks[0][key], err_msg="rank %d differs in %s" % (r, key))
    # ... and it is the single-process result, bit for bit
    GP, cand = W.model_and_candidates(sh)
    dc = DeviceArray.from_host(cand)
    one = sweep(GP, dc, acq='ei', xi=.3, native=True)
    sw = ranks[0]["sweep"]
    assert sw[0] == one["best_val"] and int(sw[1]) == one["best_idx"]
    owner = [r for r in range(world) if ranks[r]["start"] <= one["best_idx"] < ranks[r]["stop"]][0]
    assert int(sw[2]) == owner
    if world == 8:
        # eight uneven blocks; seven of the eight ranks lose the exchange in every round, and all of them know who won
        sizes = sorted(int(ranks[r]["stop"]) - int(ranks[r]["start"]) for r in range(world))
        assert sizes[0] + 1 == sizes[-1] and sum(sizes) == sh["M"]
        assert sum(1 for r in range(world) if int(ranks[r]["local_idx"]) == one["best_idx"]) == 1
    np.testing.assert_array_equal(sw[3:], cand[one["best_idx"]])
    trace = []
    gal = np.array(fastUCBGallery(GP, [[0., 1.]] * sh["D"], sh["picks"], candidates=dc, maxiter=sh["maxiter"], trace=trace))
    np.testing.assert_array_equal(ranks[0]["gallery"], gal)
    # the per-shard machinery was the kept state (two-part first sweep, refreshes afterwards), with GLOBAL indices in the
    # exchange, and at least one round was decided by a candidate of a block that does not start at row 0 or by DIRECT on every rank alike
    for r in range(world):
        k = list(ranks[r]["kernels"])
        assert k[0] == "sweep2_kernel<part>" and all(x == "sweep2_rank1_kernel" for x in k[1:]), (r, k)
        np.testing.assert_array_equal(ranks[r]["sources"], np.array([t["source"] for t in trace]))
        np.testing.assert_array_equal(ranks[r]["sweep_idx"], np.array([t["sweep_idx"] for t in trace], dtype=np.int64))
    assert np.all(ranks[0]["tiles"][:, 1] < ranks[0]["tiles"][:, 0])            # pruning was active on the shard
    X, Y, th, nz = W.grid_problem(sh)
    vals, am = nlml_grid(GaussianKernel_ard, th, X, Y, noise=nz)
    np.testing.assert_array_equal(ranks[0]["nlml"], vals)
    assert int(ranks[0]["argmin"]) == am and np.isnan(vals[sh["bad"]]) and np.isfinite(vals).sum() >= 2
    # (the not-PD theta sits in a block that is not rank 0's: its NaN must not leak into the other ranks' values through the sum)
    assert shard_bounds(sh["T"], world, 0)[1] <= sh["bad"]


def test_a_failing_rank_fails_the_launch(launch_ranks):
    """rank 1 exits non-zero right away; rank 0, waiting for it, is terminated by the launcher -- no hang, both codes reported"""
    code = ("import os, sys, time\n"
            "if os.environ['RANK'] == '1': sys.exit(3)\n"
            "time.sleep(600)\n")
    res = launch_ranks(["-c", code], 2, timeout=60)
    assert res["rc"][1] == 3 and res["rc"][0] not in (0, None), res["rc"]
    assert res["seconds"] < 30


def test_bench_line_with_two_ranks_on_one_device(launch_ranks):
    """bench.py --gpus 2 as the driver runs it (its own launcher, fresh rank processes, barriers, max-over-ranks timing, the
    candidate-sharded sweep with one arg-max exchange per step, the sharded C3 gallery and C5 grid of the `configs` block) on a box with ONE
    GPU: IBO_BENCH_ONE_DEVICE=1 puts both ranks on device 0 and the exchange on the socket transport.  The numbers mean nothing (the ranks
    share the GPU); the line's structure, the agreement of the ranks and the sharded configs' results are what is checked."""
    import json
    env = {"IBO_BENCH_ONE_DEVICE": "1"}
    res = launch_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], 1, env=env,
                       timeout=900, rank_env=False)
    assert res["rc"] == [0], res["err"][0][-3000:]
    lines = [l for l in res["out"][0].splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_nranks"] == 2 and j["transport"].startswith("socket") and j["launcher"] == "self (bench.py children)"
    assert j["scaling"] == "weak" and j["value"] > 0 and j["steps"] == 2
    c = j["configs"]
    assert c["c3_gallery8"]["min_pairwise_distance"] > 0.5 and c["c5_nlml_grid"]["n_not_pd"] == 0 and c["c3_shard_sweep"]["value"] > 0
    # the same global arg-max as ONE process sweeping the two ranks' candidates: bench.py --gpus 1 over 2^21 candidates is not a config, so
    # the check is on the index range and the value's sign here; bit-equality of sharded and single-process results is the test above
    assert 0 <= j["best"]["index"] < 2 * (1 << 20)


def test_bench_line_with_eight_ranks_on_one_device(launch_ranks):
    """The exact command line the driver issues on the 8-GPU node -- `python bench.py --gpus 8 --steps 2 --warmup 1` -- minus RCCL:
    IBO_BENCH_ONE_DEVICE=1 puts the eight rank processes on device 0 and the exchange on the socket transport.  Eight slots in every
    exchange, seven of eight ranks losing each one, the C3 gallery over eight 2^19-candidate shards (= BASELINE configs[2]'s 4M candidates)
    and the C5 grid with 64 theta-points per rank (= configs[4]'s 512) of the `configs` block.  Numbers mean nothing (the ranks share the GPU)."""
    import json
    env = {"IBO_BENCH_ONE_DEVICE": "1"}
    res = launch_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], 1, env=env,
                       timeout=1500, rank_env=False)
    assert res["rc"] == [0], res["err"][0][-3000:]
    lines = [l for l in res["out"][0].splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["rccl_nranks"] == 8 and j["transport"].startswith("socket") and j["launcher"] == "self (bench.py children)"
    assert j["scaling"] == "weak" and j["value"] > 0 and j["steps"] == 2
    assert 0 <= j["best"]["index"] < 8 * (1 << 20)
    c = j["configs"]
    assert c["c3_gallery8"]["min_pairwise_distance"] > 0.5 and c["c3_shard_sweep"]["value"] > 0
    assert "x8" in c["c3_shard_sweep"]["workload"] and "x8" in c["c5_nlml_grid"]["workload"]
    assert c["c5_nlml_grid"]["n_not_pd"] == 0 and 0 <= c["c5_nlml_grid"]["argmin"] < 512
