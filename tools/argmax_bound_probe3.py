"""Multi-level pruning of the exact arg-max (CPU, numpy): the variance from a PREFIX of W's rows bounds the true one from above, so
   EI / UCB from it bounds a candidate's value; level l covers rows [h_{l-1}, h_l) and runs only for the 32-candidate tiles whose bound
   still reaches the best exact value v* (obtained by completing the top 3 % of the level-1 ranking).  Reports, for several splits,
   the fraction of tiles surviving each level and the MFMA work relative to one full sweep (W is triangular: rows [a, b) cost
   (b^2 - a^2) / N^2).   python3 tools/argmax_bound_probe3.py [N] [M] [D] [kernel se|m5] [acq ei|ucb] [xi]"""
import sys
import numpy as np
from math import erf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
D = int(sys.argv[3]) if len(sys.argv) > 3 else 8
kern = sys.argv[4] if len(sys.argv) > 4 else "m5"
acq = sys.argv[5] if len(sys.argv) > 5 else "ei"
xi = float(sys.argv[6]) if len(sys.argv) > 6 else 0.3
noise = 0.1
rs = np.random.RandomState(3)
X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
ls = 0.3 if D <= 4 else 0.5
def K(A, B):
    d2 = np.maximum((A * A).sum(1)[:, None] + (B * B).sum(1)[None, :] - 2 * A @ B.T, 0) / ls ** 2
    if kern == "se": return np.exp(-0.5 * d2)
    r = np.sqrt(5 * d2); return (1 + r + r * r / 3) * np.exp(-r)
R = K(X, X); R[np.diag_indices(N)] = 1 + noise
L = np.linalg.cholesky(R); W = np.linalg.inv(L)
C = np.random.RandomState(103).rand(M, D)
Ks = K(C, X)
V = Ks @ W.T
mu = Ks @ (W.T @ (W @ Y))
V2 = np.cumsum(V * V, axis=1)
ymax = Y.max()
erfv = np.vectorize(erf)
def value(q):
    s = np.sqrt(np.clip(1 + noise - q, 1e-8, 10))
    if acq == "ucb": return mu + xi * s
    z = (mu - ymax - xi) / s
    return (mu - ymax - xi) * 0.5 * (1 + erfv(z / np.sqrt(2))) + s * np.exp(-z * z / 2) / np.sqrt(2 * np.pi)
exact = value(V2[:, -1])
nt = M // 32
print("N=%d M=%d D=%d %s %s(%g): best value %.6g at %d" % (N, M, D, kern, acq, xi, exact.max(), exact.argmax()))
def tile_max(v): return v.reshape(nt, 32).max(1)
for splits in ([N // 2], [N // 4, N // 2], [N // 8, N // 4, N // 2], [N // 4, N // 2, 3 * N // 4], [3 * N // 8, 5 * N // 8], [N // 8, 3 * N // 8], [N // 4], [N // 8, N // 2]):
    hs = [h // 128 * 128 for h in splits]
    b1 = tile_max(value(V2[:, hs[0] - 1]))
    top = np.argsort(-b1)[:max(1, nt * 3 // 100)]
    vstar = tile_max(exact)[top].max()
    work = (hs[0] / N) ** 2 + len(top) / nt * (1 - (hs[0] / N) ** 2)
    alive = b1 >= vstar
    alive[top] = False                                  # (already complete)
    fr = []
    prev = hs[0]
    for h in hs[1:] + [N]:
        fr.append(alive.mean())
        work += alive.mean() * ((h / N) ** 2 - (prev / N) ** 2)
        if h < N:
            alive &= tile_max(value(V2[:, h - 1])) >= vstar
        prev = h
    print("  splits %-18s v* is the true max: %-5s  tiles alive entering each later level: %s   MFMA work %.1f %% of a full sweep" %
          (hs, vstar == exact.max(), " ".join("%.2f %%" % (100 * f) for f in fr), 100 * work))
