"""
The N>1 arg-max exchange on CPU: two processes, gloo backend, the same slot
protocol RCCL runs on the GPUs (ibo_amd/multigpu.py, csrc/comm.hip).
"""
import os
import socket
import sys

import multiprocessing as mp          # torch is imported only inside the spawned workers (tests/gloo_transport.py)

import numpy as np

from conftest import ROOT

TESTS = os.path.join(ROOT, "tests")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, TESTS]
    from gloo_transport import GlooArgmax, init_gloo
    dist = init_gloo(rank, world, port)
    from ibo_amd.multigpu import shard_bounds
    comm = GlooArgmax()
    # a synthetic "acquisition" over a sharded candidate array: each rank scans its block
    M, D = 1001, 3
    cand = np.random.RandomState(5).rand(M, D)
    vals = np.sin(7 * cand.sum(1))
    vals[[17, 600, 900]] = 2.0                         # three-way tie across ranks -> index 17 must win
    a, b = shard_bounds(M, world, rank)
    li = int(np.argmax(vals[a:b]))
    out = [comm.argmax(vals[a + li], a + li, cand[a + li])]
    # a rank with nothing admissible (everything excluded) must not win
    out.append(comm.argmax(float('nan') if rank == 0 else -5.0 - rank, -1 if rank == 0 else 10 + rank, [0.] * D))
    out.append(comm.argmax(0.0, -1, []))
    q.put((rank, [(v, i, list(p), r) for v, i, p, r in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_argmax_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1]                            # identical outcome on every rank
    cand = np.random.RandomState(5).rand(1001, 3)
    v, i, p, r = res[0][0]
    assert (v, i, r) == (2.0, 17, 0) and np.allclose(p, cand[17])
    v, i, p, r = res[0][1]
    assert (v, i, r) == (-6.0, 11, 1)
    assert res[0][2][1] == -1 and res[0][2][3] == -1


def _tie_worker(rank, world, port, q):
    sys.path[:0] = [ROOT, TESTS]
    from gloo_transport import GlooArgmax, init_gloo
    dist = init_gloo(rank, world, port)
    from ibo_amd.multigpu import shard_bounds
    comm = GlooArgmax()
    M = 30
    vals = np.zeros(M)
    vals[[4, 14, 24]] = 1.5                            # one maximiser in EACH of the three shards
    a, b = shard_bounds(M, world, rank)
    li = int(np.argmax(vals[a:b]))
    out = [comm.argmax(vals[a + li], a + li, [float(a + li)])]
    # global indices beyond 2^31 (and beyond 2^32) travel exactly in the fp64 slot
    base = (1 << 33) + 12345
    out.append(comm.argmax(1.0, base + (world - 1 - rank), [float(rank)]))
    out.append(comm.argmax(float(rank == 1), (1 << 52) + rank, []))
    q.put((rank, [(v, int(i), list(p), r) for v, i, p, r in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_three_shard_tie_and_large_indices_gloo():
    """the same value in all three shards: the lowest GLOBAL index wins on every rank (numpy.argmax
    order); indices above 2^31 survive the all-reduce(sum) exactly"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tie_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1] == res[2]
    assert res[0][0] == (1.5, 4, [4.0], 0)
    base = (1 << 33) + 12345
    assert res[0][1] == (1.0, base, [2.0], 2)          # rank 2 holds the lowest index of the tie
    assert res[0][2] == (1.0, (1 << 52) + 1, [], 1)


def _id_worker(rank, world, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_PORT"] = "45678"
    from ibo_amd import multigpu
    multigpu.RcclArgmax.unique_id = staticmethod(lambda: bytes(range(128)))     # no RCCL on the CPU box
    if rank != 0:
        import time
        time.sleep(0.3)                                # rank 0 may well be first ...
    uid, path = multigpu.exchange_unique_id(world, rank, timeout_s=30)
    q.put((rank, uid, path))


def test_unique_id_file_rendezvous():
    """bench.py's torch-free rendezvous: rank 0 publishes the RCCL id in a private per-user directory,
    keyed by the launcher pid, the port and torch-elastic's attempt nonce"""
    world = 3
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_id_worker, args=(r, world, q)) for r in range(world)]
    for p in reversed(procs):                          # ... or last
        p.start()
    res = [q.get(timeout=60) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(uid == bytes(range(128)) for _, uid, _ in res)
    assert len(set(path for _, _, path in res)) == 1
    st = os.stat(res[0][2])
    assert st.st_mode & 0o077 == 0 and os.stat(os.path.dirname(res[0][2])).st_mode & 0o077 == 0
    os.unlink(res[0][2])


def test_rendezvous_name_carries_the_attempt_nonce(monkeypatch, tmp_path):
    """a restarted torch-elastic attempt must not read the id file of the attempt before it"""
    from ibo_amd import multigpu
    monkeypatch.setenv("IBO_COMM_DIR", str(tmp_path / "rdv"))
    monkeypatch.delenv("IBO_COMM_ID_FILE", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29500")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/7")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    a = multigpu._rendezvous_path()
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    b = multigpu._rendezvous_path()
    assert a != b and os.path.dirname(a) == str(tmp_path / "rdv") and "/" not in os.path.basename(a)
    os.chmod(str(tmp_path / "rdv"), 0o755)             # somebody else could write here: refuse
    import pytest
    with pytest.raises(RuntimeError):
        multigpu._rendezvous_path()
    monkeypatch.setenv("IBO_COMM_ID_FILE", "/somewhere/explicit")
    assert multigpu._rendezvous_path() == "/somewhere/explicit"


def _nlml_worker(rank, world, port, q):
    sys.path[:0] = [ROOT, TESTS]
    from gloo_transport import GlooArgmax, init_gloo
    dist = init_gloo(rank, world, port)
    from ibo_amd.multigpu import sharded_nlml_grid
    comm = GlooArgmax()
    thetas = np.random.RandomState(9).rand(11, 3) + .1
    f = lambda th: np.where(th[:, 0] > .95, np.nan, np.sum((th - .5) ** 2, axis=1))     # NaN = "not PD" slots
    vals, am = sharded_nlml_grid(None, thetas, None, None, comm, local_eval=f)
    q.put((rank, vals.tolist(), am))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_nlml_grid_gather_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nlml_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (v, a)) for r, v, a in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    thetas = np.random.RandomState(9).rand(11, 3) + .1
    ref = np.where(thetas[:, 0] > .95, np.nan, np.sum((thetas - .5) ** 2, axis=1))
    for r in range(world):
        np.testing.assert_array_equal(np.array(res[r][0]), ref)
        assert res[r][1] == int(np.nanargmin(ref))


def test_socket_transport_three_ranks_in_threads():
    """ibo_amd.multigpu.SocketComm (the transport of tests/test_gpu_two_ranks.py): sum in rank order, the slot protocol's
    tie rule (lowest global index), a rank with nothing admissible, indices beyond 2^32 -- three ranks as three threads"""
    import tempfile
    import threading
    from ibo_amd.multigpu import SocketComm
    world = 3
    res = {}
    with tempfile.TemporaryDirectory() as d:
        addr = os.path.join(d, "sock")

        def run(rank):
            c = SocketComm(world, rank, addr, timeout_s=60)
            out = [c.allreduce_sum([rank + 1.0, 0.25, -rank])]
            out.append(c.argmax(2.0, (1 << 33) + 10 - rank, [float(rank), 7.0]))                 # a three-way tie: the lowest index wins
            out.append(c.argmax(float('nan') if rank == 2 else -1.0 - rank, -1 if rank == 2 else rank, [0.0]))
            out.append(c.argmax(0.0, -1, []))                                                     # nobody has anything
            c.barrier()
            assert c.nranks() == world
            c.close()
            res[rank] = out
        ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(120)
        assert not os.path.exists(addr)                    # rank 0 removed its socket
    assert sorted(res) == [0, 1, 2]
    for r in range(world):
        np.testing.assert_array_equal(res[r][0], [6.0, 0.75, -3.0])
        v, i, p, who = res[r][1]
        assert (v, i, who) == (2.0, (1 << 33) + 8, 2) and list(p) == [2.0, 7.0]
        v, i, p, who = res[r][2]
        assert (v, i, who) == (-1.0, 0, 0)
        assert res[r][3][1] == -1 and res[r][3][3] == -1


def test_rank_launcher_starts_ranks_and_reports_failures(launch_ranks):
    """tests/rank_launcher.py (the GPU-clean parent of the GPU tests' rank processes): RANK / WORLD_SIZE per child, outputs
    back, and a failing rank takes the launch down instead of hanging it"""
    ok = launch_ranks(["-c", "import os; print(os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'])"], 3, timeout=60)
    assert ok["rc"] == [0, 0, 0] and [o.split() for o in ok["out"]] == [[str(r), "3", str(r)] for r in range(3)]
    bad = launch_ranks(["-c", "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(600)\n"], 2, timeout=60)
    assert bad["rc"][1] == 3 and bad["rc"][0] not in (0, None) and bad["seconds"] < 30
