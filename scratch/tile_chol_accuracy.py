"""Backward error || A - L L^T || / || A || of a 128 x 128 tile Cholesky built the way potrf_tile_kernel builds it (16 x 16 blocks,
the panel below a diagonal block as a PRODUCT with the explicit inverse of that block) against LAPACK's, and with one step of
refinement in that product (CPU, numpy: the algorithm, not the kernel)."""
import numpy as np, scipy.linalg as sla, sys
def m52(r): return (1 + r + r * r / 3) * np.exp(-r)
def tile_chol(A, refine):
    A = A.copy(); n = A.shape[0]
    for j0 in range(0, n, 16):
        D = np.linalg.cholesky(A[j0:j0+16, j0:j0+16])
        A[j0:j0+16, j0:j0+16] = D
        if j0 + 16 < n:
            Dinv = sla.solve_triangular(D, np.eye(16), lower=True)
            B = A[j0+16:, j0:j0+16]
            X = B @ Dinv.T
            if refine:
                X = X + (B - X @ D.T) @ Dinv.T
            A[j0+16:, j0:j0+16] = X
            A[j0+16:, j0+16:] -= X @ X.T
    return np.tril(A)
rng = np.random.default_rng(0)
print("case                          cond(A)    LAPACK     product    refined   | max cond(16-blk of L)")
for name, pts, noise in [("grid 1-D dense, noise 1e-8", np.linspace(-1, 1, 128)[:, None], 1e-8),
                          ("grid 1-D dense, noise 1e-10", np.linspace(-1, 1, 128)[:, None], 1e-10),
                          ("scattered 2-D, noise 1e-8", rng.uniform(-1, 1, (128, 2)), 1e-8),
                          ("scattered 2-D, noise 1e-6", rng.uniform(-1, 1, (128, 2)), 1e-6),
                          ("clustered 2-D, noise 1e-9", 0.05 * rng.standard_normal((128, 2)), 1e-9)]:
    a = np.sqrt(5.0)
    K = np.ones((128, 128))
    for d in range(pts.shape[1]):
        K = K * m52(a * np.abs(pts[:, None, d] - pts[None, :, d]))
    A = 4.0 * K + noise * np.eye(128)
    Ll = np.linalg.cholesky(A)
    be = lambda L: np.abs(A - L @ L.T).max() / np.abs(A).max()
    L1, L2 = tile_chol(A, False), tile_chol(A, True)
    kb = max(np.linalg.cond(Ll[i:i+16, i:i+16]) for i in range(0, 128, 16))
    # forward error of L against a long-double Cholesky
    Al = A.astype(np.longdouble); Lx = np.zeros_like(Al)
    for j in range(128):
        Lx[j, j] = np.sqrt(Al[j, j] - (Lx[j, :j] ** 2).sum())
        Lx[j+1:, j] = (Al[j+1:, j] - Lx[j+1:, :j] @ Lx[j, :j]) / Lx[j, j]
    fe = lambda L: float(np.abs(L.astype(np.longdouble) - Lx).max() / np.abs(Lx).max())
    print(f"{name:28s} {np.linalg.cond(A):9.2e}  {be(Ll):9.2e}  {be(L1):9.2e}  {be(L2):9.2e}  | {kb:9.2e}   forward: {fe(Ll):.2e} {fe(L1):.2e} {fe(L2):.2e}")
