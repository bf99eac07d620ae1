"""
bench.py's own multi-process launcher (`python bench.py --gpus N` without torch.distributed.run),
exercised on CPU: the rank environment, the private rendezvous file, the relay of rank 0's single
JSON line, and the failure paths.  The real workers need a GPU, so the success path runs a small
stand-in rank script through the same launcher function.
"""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

RANK_SCRIPT = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
    from ibo_amd import multigpu
    multigpu.RcclArgmax.unique_id = staticmethod(lambda: bytes(range(128)))      # no RCCL on the CPU box
    uid, path = multigpu.exchange_unique_id(world, rank, timeout_s=30)
    assert uid == bytes(range(128)) and path == os.environ["IBO_COMM_ID_FILE"]
    assert os.stat(os.path.dirname(path)).st_mode & 0o077 == 0
    mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
    if mode == "fail" and rank == 1:
        sys.exit(7)
    if mode == "fail":
        import time; time.sleep(600)                                             # must be stopped by the launcher
    print("rank %%d noise" %% rank, file=sys.stderr)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "id_dir": os.path.dirname(path)}))
    else:
        print("not the JSON line")                                               # other ranks' stdout never reaches ours
''') % ROOT


def _run_launcher(tmp_path, n, mode):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    code = ("import sys; sys.path.insert(0, %r); import bench; rc = bench.self_launch(%d, [%r], timeout_s=120, script=%r); "
            "assert 'ibo_amd' not in sys.modules, 'the launcher process must not load the GPU library'; sys.exit(rc)"
            % (ROOT, n, mode, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "IBO_COMM_ID_FILE")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)


def test_self_launcher_relays_rank0_json(tmp_path):
    p = _run_launcher(tmp_path, 3, "ok")
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 3
    assert not os.path.exists(j["id_dir"])                 # the private rendezvous directory is removed
    assert "not the JSON line" in p.stderr


def test_self_launcher_fails_when_a_rank_fails(tmp_path):
    p = _run_launcher(tmp_path, 2, "fail")
    assert p.returncode == 7, (p.returncode, p.stderr)
    assert p.stdout.strip() == ""
    assert "rank 1 exited with status 7" in p.stderr


def test_bench_without_gpu_fails_loudly():
    """the real thing on a box without a GPU: both ranks report the missing device, the launcher exits non-zero
    and prints no JSON (there is no CPU fallback to time)"""
    import ctypes
    import ibo_amd
    n = ctypes.c_int(-1)
    ibo_amd._lib.lib.ibo_device_count(ctypes.byref(n))
    if n.value > 0:
        import pytest
        pytest.skip("a GPU is visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "IBO_COMM_ID_FILE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "no HIP device" in p.stderr or "error 5" in p.stderr
