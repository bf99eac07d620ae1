"""C5 (N = 4096, D = 16, 64 theta) wall time, A/B over ibo_set_option settings in one process.
python3 tools/c5_only.py [key=value[,key=value] ...]   e.g.  nlml_batch=32 nlml_batch=64 nlml_batch=16"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibo_amd import _lib
from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard
from ibo_amd.gaussianprocess.trainhyper import nlml_grid
rs = np.random.RandomState(5); X = rs.rand(4096, 16); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(4096)
th = np.exp(np.random.RandomState(105).uniform(np.log(.1), np.log(3), size=(64, 16)))
settings = sys.argv[1:] or ["nlml_batch=0"]
DEFAULTS = dict(nlml_batch=0, chol_left=1)
ref = None
res = {s: [] for s in settings}
for rnd in range(4):
    for st in settings:
        for k, v in DEFAULTS.items():                   # options are sticky: every setting starts from the defaults
            _lib.check(_lib.lib.ibo_set_option(k.encode(), v))
        for kv in st.split(","):
            k, v = kv.split("="); _lib.check(_lib.lib.ibo_set_option(k.encode(), int(v)))
        t0 = time.perf_counter(); vals = np.array(nlml_grid(GaussianKernel_ard, th, X, Y, noise=1e-3)[0]); dt = (time.perf_counter() - t0) * 1e3
        if ref is None: ref = vals
        same = np.array_equal(vals, ref)
        if rnd: res[st].append(dt)
        if rnd == 1: print("%-28s same values as the first setting: %s" % (st, same), flush=True)
import hashlib
print("values sha1", hashlib.sha1(ref.tobytes()).hexdigest()[:16])
for st in settings:
    print("%-28s %.2f ms (min %.2f) = %.3f ms/theta" % (st, np.median(res[st]), min(res[st]), np.median(res[st]) / 64))
