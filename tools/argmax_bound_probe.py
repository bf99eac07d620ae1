"""How much of a sweep an exact arg-max-only mode could skip: after p of the W row panels q_p <= q bounds
   sigma^2 <= 1 + noise - q_p, EI is increasing in sigma, so a candidate whose EI(mu, sigma_ub) is below the incumbent can stop.
   CPU estimate on C2-shaped synthetic data (numpy only, no GPU): fraction of (tile, panel) MFMA work a tile-granular skip saves,
   given the best possible incumbent (the true maximum).
   python3 tools/argmax_bound_probe.py [N] [M]"""
import sys
import numpy as np
from scipy.special import erf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
D, noise, xi = 4, 0.01, 0.01
rs = np.random.RandomState(3)
def ei(mu, s, ymax):
    z = (mu - ymax - xi) / s
    return (mu - ymax - xi) * 0.5 * (1 + erf(z / np.sqrt(2))) + s * np.exp(-0.5 * z * z) / np.sqrt(2 * np.pi)
for name, order in (("data order as given (random)", None), ("rows sorted along the first coordinate", 0)):
    X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
    if order is not None:
        p = np.argsort(X[:, order]); X, Y = X[p], Y[p]
    ls = 0.3
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    R = np.exp(-0.5 * d2 / ls ** 2) + noise * np.eye(N)
    L = np.linalg.cholesky(R); W = np.linalg.inv(L)
    C = rs.rand(M, D)
    Ks = np.exp(-0.5 * ((C[:, None, :] - X[None, :, :]) ** 2).sum(-1) / ls ** 2)          # M x N
    V = Ks @ W.T                                                                        # M x N: (W k*)_i per candidate
    q_rows = V * V
    mu = Ks @ (W.T @ (W @ Y))
    q = q_rows.sum(1); s2 = np.maximum(1 + noise - q, 1e-12)
    ymax = Y.max(); val = ei(mu, np.sqrt(s2), ymax); best = val.max()
    # panels of 64 rows; work of panel g (lower-triangular W): proportional to g + 1
    npan = N // 64
    qp = np.cumsum(q_rows.reshape(M, npan, 64).sum(2), axis=1)                           # q after panel g
    ub = ei(mu[:, None], np.sqrt(np.maximum(1 + noise - qp, 1e-12)), ymax)              # bound after panel g
    prunable = ub < best                                                                 # candidate could stop after panel g
    tile = prunable.reshape(M // 32, 32, npan).all(1)                                    # a 32-candidate tile stops when all of it can
    first = np.where(tile.any(1), tile.argmax(1), npan - 1)                              # last panel the tile must run
    w = np.arange(1, npan + 1, dtype=float)
    done = np.array([w[:f + 1].sum() for f in first]); total = w.sum()
    print("%-40s N=%d M=%d: tiles that stop early %.1f %%, MFMA work skipped %.1f %% (median q / (1+noise) = %.2f)" %
          (name, N, M, 100 * (first < npan - 1).mean(), 100 * (1 - done.mean() / total), np.median(q) / (1 + noise)))
