"""CPU emulation: how much accuracy does the explicit-tile-inverse TRSM cost, in the factorisation and in
the forward substitution, and what does ONE refinement step per tile product (X = X0 + (A - X0 L^T) Linv^T)
recover?  Blocked right-looking Cholesky on 128-tiles exactly as potrf_blocked does it, in NumPy."""
import sys, os, time
import numpy as np, scipy.linalg
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
from linpde_gp_amd import problems
from oracle import workloads as owl, gp as ogp

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 48
wl = problems.poisson_2d(n_side, m_side=24)
blocks = owl.blocks_of(wl)
t = time.time(); G = ogp.gram(wl.kernel, blocks); r = ogp.residual(blocks); K = ogp.cross_cov(wl.kernel, blocks, wl.Xtest)
print("N", G.shape[0], "assembly", time.time() - t)
N = G.shape[0]; TB = 128
pad = (-N) % TB
Gp = np.eye(N + pad); Gp[:N, :N] = G
rp = np.concatenate([r, np.zeros(pad)]); Kp = np.concatenate([K.T, np.zeros((pad, K.shape[0]))])
T = (N + pad) // TB
tri = lambda L, B: scipy.linalg.solve_triangular(L, B, lower=True, check_finite=False)

def factor(refine_f, exact_f=False):
    A = Gp.copy(); Linv = []
    for j in range(T):
        s = slice(j * TB, (j + 1) * TB)
        Ljj = np.linalg.cholesky(A[s, s]); A[s, s] = Ljj
        Li = tri(Ljj, np.eye(TB)); Linv.append(Li)
        if j + 1 < T:
            b = slice((j + 1) * TB, None)
            Ab = A[b, s]
            if exact_f:
                X = tri(Ljj, Ab.T).T
            else:
                X = Ab @ Li.T
                if refine_f:
                    X = X + (Ab - X @ Ljj.T) @ Li.T
            A[b, s] = X
            A[b, b] -= X @ X.T
    return np.tril(A), Linv

def fwd(L, Linv, B, refine_s, exact_s=False):
    V = B.copy()
    for j in range(T):
        s = slice(j * TB, (j + 1) * TB)
        if exact_s:
            V[s] = tri(L[s, s], V[s])
        else:
            X = Linv[j] @ V[s]
            if refine_s:
                X = X + Linv[j] @ (V[s] - L[s, s] @ X)
            V[s] = X
        if j + 1 < T:
            b = slice((j + 1) * TB, None)
            V[b] -= L[b, s] @ V[s]
    return V

# truth: refined solve with long-double residuals
c = scipy.linalg.cholesky(G, lower=True)
w = scipy.linalg.cho_solve((c, True), r); Gl = G.astype(np.longdouble)
for _ in range(3):
    w = w + scipy.linalg.cho_solve((c, True), np.asarray(r.astype(np.longdouble) - Gl @ w.astype(np.longdouble), dtype=np.double))
m_true = K @ w
m_lapack = K @ scipy.linalg.cho_solve((c, True), r)
Vl = tri(c, K.T); v_lapack = 4.0 - np.sum(Vl * Vl, 0)
rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
print(f"LAPACK vs refined truth: mean {rel(m_lapack, m_true):.2e}")
conds = []
for name, rf, ef, rs, es in [("tile inverses, no refinement (round 1)", 0, 0, 0, 0), ("refined factorisation only", 1, 0, 0, 0),
                             ("refined substitution only", 0, 0, 1, 0), ("refined both", 1, 0, 1, 0), ("exact tile substitution both", 0, 1, 0, 1)]:
    L, Linv = factor(rf, ef)
    V = fwd(L, Linv, np.column_stack([Kp, rp]), rs, es)
    z = V[:, -1]; V = V[:, :-1]
    mean = V.T @ z; var = 4.0 - np.sum(V * V, 0)
    print(f"{name:42s}: mean vs truth {rel(mean, m_true):.2e}  vs LAPACK {rel(mean, m_lapack):.2e};  var vs LAPACK {rel(var, v_lapack):.2e}")
L, Linv = factor(0)
print("cond of the diagonal tiles of L: max %.2e median %.2e" % (max(np.linalg.cond(L[j*TB:(j+1)*TB, j*TB:(j+1)*TB]) for j in range(T)),
      np.median([np.linalg.cond(L[j*TB:(j+1)*TB, j*TB:(j+1)*TB]) for j in range(T)])))
