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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


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
