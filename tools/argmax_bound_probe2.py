"""Two-panel pruning for the gallery (CPU, numpy): rows [0, N/2) of W first (a quarter of the MFMA work), then a tile of 32
   candidates runs the second half only if some candidate's bound UCB(mu, sigma_ub) reaches the best value known.  Simulates
   fastUCBGallery's rounds (hallucinated observation at the mean: q += z^2, mu unchanged): how many tiles need their second half, per round.
   python3 tools/argmax_bound_probe2.py [N] [M] [D]"""
import sys
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
D = int(sys.argv[3]) if len(sys.argv) > 3 else 4
noise, kappa = 0.01, 2.0
rs = np.random.RandomState(5)
X = rs.rand(N, D); Y = np.sin(3 * X.sum(1)) + 0.01 * rs.randn(N)
ls = 0.3 if D <= 4 else 0.5
def K(A, B): return np.exp(-0.5 * ((A[:, None, :] - B[None, :, :]) ** 2).sum(-1) / ls ** 2)
R = K(X, X) + noise * np.eye(N)
L = np.linalg.cholesky(R); W = np.linalg.inv(L)
C = rs.rand(M, D)
Ks = K(C, X)
V = Ks @ W.T
h = N // 2
q0 = (V[:, :h] ** 2).sum(1); q1 = (V[:, h:] ** 2).sum(1)
mu = Ks @ (W.T @ (W @ Y))
ucb = lambda q: mu + kappa * np.sqrt(np.maximum(1 + noise - q, 1e-12))
complete = np.zeros(M // 32, bool)
zsum = np.zeros(M)
Xc, Wc = X.copy(), W.copy()
total_second = 0
for rnd in range(8):
    # exact values where complete, bounds elsewhere
    cand_complete = np.repeat(complete, 32)
    while True:
        exact = np.where(cand_complete, ucb(q0 + q1 + zsum), -np.inf)
        bound = ucb(q0 + zsum)
        best = exact.max()
        need = (~complete) & (bound.reshape(-1, 32) >= best).any(1)
        if not need.any(): break
        # (the kernel would finish the needed tiles in index order, the running best rising as it goes: emulate one tile at a time in
        # descending order of their best bound -- what a second launch over the compacted list achieves)
        order = np.argsort(-np.where(need, bound.reshape(-1, 32).max(1), -np.inf))
        t = order[0]
        complete[t] = True; cand_complete[32 * t:32 * t + 32] = True; total_second += 1
    i = int(np.argmax(exact))
    print("round %d: winner %6d  value %.6f  tiles with second half so far %5d of %d (%.2f %%)" % (rnd + 1, i, exact[i], complete.sum(), M // 32, 100 * complete.mean()))
    # hallucinate at the winner: new row of W: z = W k, d = sqrt(1 + noise - |z|^2); candidate side: z_c = (k*(c, x) - z . (W k*_c)) / d
    x = C[i:i + 1]
    kx = K(Xc, x)[:, 0]
    z = Wc @ kx; d = np.sqrt(1 + noise - z @ z)
    Vc = K(C, Xc) @ Wc.T                                   # (CPU shortcut: recompute V for the grown model)
    zc = (K(C, x)[:, 0] - Vc @ z) / d
    zsum += zc ** 2
    wrow = np.concatenate([-(Wc.T @ z) / d, [1 / d]])
    Wc = np.block([[Wc, np.zeros((len(Wc), 1))], [wrow[None, :]]]); Xc = np.vstack([Xc, x])
print("MFMA work relative to eight unpruned first sweeps' one: first halves 25 %% + second halves %.1f %% = %.1f %% of ONE full sweep" %
      (75.0 * complete.mean(), 25 + 75.0 * complete.mean()))
