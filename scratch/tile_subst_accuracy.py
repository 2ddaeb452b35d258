"""Backward error of three ways to solve X L^T = A against a 128x128 Cholesky tile (CPU, numpy):
 (i) product with the explicit tile inverse, (ii) the same refined once (what tile_solve_kernel does),
 (iii) block substitution over 16x16 blocks with the inverses of the DIAGONAL blocks only.
Tiles come from a right-looking tile Cholesky of a Matern-5/2 product-kernel Gram on a 2-D grid (+1e-8 noise)."""
import numpy as np, scipy.linalg as sla, sys
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 40
T = 128
def m52(r): return (1 + r + r * r / 3) * np.exp(-r)
g = np.linspace(-1, 1, n1)
X = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
a = np.sqrt(5.0)
K = 4.0 * m52(a * np.abs(X[:, None, 0] - X[None, :, 0])) * m52(a * np.abs(X[:, None, 1] - X[None, :, 1]))
K[np.diag_indices_from(K)] += 1e-8
n = (K.shape[0] // T) * T
K = K[:n, :n].copy()
print("n", n, "cond2 ~", np.linalg.cond(K))
A = K.copy()
res = []
for j in range(n // T):
    s = slice(j * T, (j + 1) * T)
    L = np.linalg.cholesky(A[s, s])
    Linv = sla.solve_triangular(L, np.eye(T), lower=True)
    B = A[(j + 1) * T:, s]
    if B.shape[0] == 0: break
    Xex = sla.solve_triangular(L.astype(np.longdouble).astype(np.float64), B.T, lower=True).T   # LAPACK substitution
    # exact-ish: long double substitution
    Ll = L.astype(np.longdouble); Bl = B.astype(np.longdouble)
    Xl = np.zeros_like(Bl)
    for c in range(T):
        Xl[:, c] = (Bl[:, c] - Xl[:, :c] @ Ll[c, :c]) / Ll[c, c]
    def err(Xc): return float(np.max(np.abs(Xc.astype(np.longdouble) - Xl)) / np.max(np.abs(Xl)))
    X1 = B @ Linv.T
    R = B - X1 @ L.T
    X2 = X1 + R @ Linv.T
    X3 = np.zeros_like(B); W = B.copy()
    for c in range(T // 16):
        cs = slice(16 * c, 16 * c + 16)
        Dinv = Linv[cs, cs]                      # == inverse of L[cs, cs]
        X3[:, cs] = W[:, cs] @ Dinv.T
        rest = slice(16 * c + 16, T)
        W[:, rest] -= X3[:, cs] @ L[rest, cs].T
    # (iv) block substitution, diagonal-block solve refined once
    X4 = np.zeros_like(B); W = B.copy()
    for c in range(T // 16):
        cs = slice(16 * c, 16 * c + 16)
        Dinv = Linv[cs, cs]; D = L[cs, cs]
        x = W[:, cs] @ Dinv.T
        x = x + (W[:, cs] - x @ D.T) @ Dinv.T
        X4[:, cs] = x
        rest = slice(16 * c + 16, T)
        W[:, rest] -= x @ L[rest, cs].T
    dr = L.diagonal().max() / L.diagonal().min(); nm = np.abs(Linv).sum(1).max() * np.abs(L).sum(1).max()
    kt = np.linalg.cond(L); kb = max(np.linalg.cond(L[16*c:16*c+16, 16*c:16*c+16]) for c in range(8))
    res.append((j, kt, kb, dr, nm, err(Xex), err(X1), err(X2), err(X3), err(X4)))
    # continue the factorisation with the LAPACK-quality panel
    A[(j + 1) * T:, s] = Xex
    A[(j + 1) * T:, (j + 1) * T:] -= Xex @ Xex.T
print(" tile  cond(Ljj)  max cond(16-blk) diag-ratio  inf-norm-cond  lapack     inv-prod   refined    blk-subst  blk-subst-refined")
for r in res: print("%4d  %9.2e  %9.2e   %9.2e  %9.2e   %9.2e  %9.2e  %9.2e  %9.2e  %9.2e" % r)
r = np.array(res)
print("max   %9.2e  %9.2e   %9.2e  %9.2e   %9.2e  %9.2e  %9.2e  %9.2e  %9.2e" % tuple(r[:, 1:].max(0)))
