"""Latin hypercube sampling (ego/utils/latinhypercube.py:27-46).

Host-side candidate generator; the RandomState draw order (per dimension:
rand(N) then shuffle) is kept so seeded samples are bit-identical to the
reference's (tests/golden/g9_lhc.npz)."""
import numpy as np
from numpy.random import RandomState


def lhcSample(bounds, N, seed=None):
    """N stratified samples in the box `bounds` (sequence of [min, max]); a
    dimension with min == max is held constant.  Returns a list of N arrays (D,)."""
    rs = RandomState(seed)
    samp = []
    for bmin, bmax in bounds:
        if bmin == bmax:
            dsamp = np.array([bmin] * N)
        else:
            dsamp = (bmax - bmin) * rs.rand(N) / N + np.arange(bmin, bmax, (bmax - bmin) / N)
        rs.shuffle(dsamp)
        samp.append(dsamp)
    return list(np.vstack(samp).T)
