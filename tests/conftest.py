import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _ensure_built():
    """The .so files are git-ignored build products: build them when a fresh checkout runs the tests
    (hipcc cross-compiles gfx950 without a GPU; gcc builds the CPU checker).  Building is not using."""
    import subprocess
    lib = os.path.join(ROOT, "ibo_amd", "libibo_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "ibo_amd", "csrc"), "-j4"])
    orc = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(orc) or (os.path.isdir("/root/reference/cpp") and
                                   not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libego.so"))):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


_LAUNCHER = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()
    # the helper that starts rank processes for the GPU tests: created NOW, while this process has not touched the GPU
    # (tests/rank_launcher.py says why); idle unless a test asks
    global _LAUNCHER
    import subprocess
    _LAUNCHER = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_launcher.py")], stdin=subprocess.PIPE,
                                 stdout=subprocess.PIPE, cwd=ROOT)


def pytest_unconfigure(config):
    global _LAUNCHER
    if _LAUNCHER is not None:
        try:
            _LAUNCHER.stdin.close()                      # EOF ends it
            _LAUNCHER.wait(10)
        except Exception:
            _LAUNCHER.kill()
        _LAUNCHER = None


@pytest.fixture(scope="session")
def launch_ranks():
    """launch_ranks(argv, world, env=None, timeout=600, rank_env=True) -> dict(rc, out, err): `world` fresh python processes running argv,
    started by a helper that never touched the GPU"""
    import json

    def launch(argv, world, env=None, timeout=600, rank_env=True):
        assert _LAUNCHER is not None and _LAUNCHER.poll() is None, "the rank launcher is not running"
        job = {"argv": list(argv), "world": int(world), "env": env or {}, "timeout": timeout, "rank_env": bool(rank_env)}
        _LAUNCHER.stdin.write((json.dumps(job) + "\n").encode())
        _LAUNCHER.stdin.flush()
        return json.loads(_LAUNCHER.stdout.readline().decode())
    return launch


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (test infrastructure; never imported by ibo_amd)."""
    from oracle import oracle as orc
    orc.lib()
    return orc


def kern_from_golden(orc, ktype, hyper):
    """Rebuild an oracle Kern from the (ktype, hyperparams) stored in a fixture."""
    kind = {0: "ard", 1: "iso", 2: "m3", 3: "m5"}[int(ktype)]
    return orc.Kern(kind, np.array(hyper, dtype=float))


def synth(seed, N, D):
    """Synthetic GP data of SURVEY 8(d): X=rand, Y=sin(3*sum X)+0.01 randn."""
    rs = np.random.RandomState(seed)
    X = rs.rand(N, D)
    Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    return X, Y
