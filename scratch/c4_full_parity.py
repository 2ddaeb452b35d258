"""c4 (2-D Poisson 256 x 256, N_tot = 66 560, M = 16 384; default) or c5 (`c5`: heat, N_tot = 33 600) on ONE GPU against the CPU oracle AT FULL SIZE.
Not part of the test suite (a quarter of an hour of host time, ~90 GB of host memory): the oracle's Gram matrix is
assembled in row chunks (the plain `oracle.gp.gram` would need ~12 matrix-sized temporaries), factored in place by LAPACK.
Prints the parity numbers with the criterion of tests/conftest.py."""
import os, sys, time
import numpy as np, scipy.linalg
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
import psutil
import linpde_gp_amd as lp
from linpde_gp_amd import problems
from oracle import workloads as owl, gp as ogp, covfuncs as ocf

if len(sys.argv) > 1 and sys.argv[1] == "c5":
    wl = problems.heat_1d()                                   # c5 at full size: N_tot = 33 600, M = 4 096
else:
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    m_side = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    wl = problems.poisson_2d(n_side, m_side=m_side)
N, M = wl.n_total, wl.Xtest.shape[0]
need = (8.0 * N * N + 2 * 8.0 * N * M) * 1.15 + 40e9
avail = psutil.virtual_memory().available
print(f"N_tot={N} M={M}: host memory needed ~{need/1e9:.0f} GB, available {avail/1e9:.0f} GB, cpus {os.cpu_count()}", flush=True)
if avail < need:
    raise SystemExit("not enough host memory for the full-size oracle")
lp.config.gram_capacity_hint = N
t0 = time.time()
u, mean, var = problems.condition_and_predict(wl)
print(f"device: {time.time() - t0:.2f} s (first call, includes allocation)", flush=True)
del u
blocks = owl.blocks_of(wl)
off = np.cumsum([0] + [b.n for b in blocks])
G = np.zeros((N, N))
t0 = time.time()
CH = 2048
for i, bi in enumerate(blocks):
    for j, bj in enumerate(blocks[:i + 1]):
        for r0 in range(0, bi.n, CH):
            r1 = min(bi.n, r0 + CH)
            c1 = bj.n if i != j else r1          # lower triangle of the diagonal block only
            G[off[i] + r0:off[i] + r1, off[j]:off[j] + c1] = ocf.LkL(wl.kernel, bi.L, bj.L, bi.X[r0:r1], bj.X[:c1])
    if bi.noise_cov is not None:
        idx = np.arange(off[i], off[i + 1])
        G[idx, idx] += float(bi.noise_cov)
print(f"oracle assembly {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
chol = scipy.linalg.cholesky(G, lower=True, overwrite_a=True, check_finite=False)
print(f"oracle dpotrf {time.time() - t0:.1f} s", flush=True)
r = ogp.residual(blocks)
w = scipy.linalg.cho_solve((chol, True), r, check_finite=False)
t0 = time.time()
K = np.empty((M, N))
for r0 in range(0, M, CH):
    K[r0:r0 + CH] = ogp.cross_cov(wl.kernel, blocks, wl.Xtest[r0:r0 + CH])
ref_mean = K @ w
Kc = K.copy()
V = scipy.linalg.solve_triangular(chol, K.T, lower=True, overwrite_b=True, check_finite=False)
kxx = float(sum(sc for sc, _ in wl.kernel))
ref_var = kxx - ogp.colsumsq(V)
var_naive = kxx - np.einsum("ij,ij->j", V, V)      # what the oracle did before: N-term sequential accumulation
print(f"oracle prediction {time.time() - t0:.1f} s", flush=True)
# The oracle's own rounding floor on the variance: the same quantity by two other fp64-valid routes from the same factor --
# (b) k(x,x) - k_x^T (K^-1 k_x) through both triangular solves, (c) route (a) with the sum of squares accumulated in long double.
W = scipy.linalg.solve_triangular(chol, V, lower=True, trans="T", check_finite=False)
var_b = kxx - np.einsum("ji,ij->j", Kc, W)
var_c = kxx - np.asarray((V.astype(np.longdouble) ** 2).sum(axis=0), dtype=float) if M * N <= 2e8 else ref_var
print(f"naive N-term accumulation vs pairwise: {np.max(np.abs(ref_var - var_naive)):.3e}; device vs naive {np.max(np.abs(var - var_naive)):.3e}")
print(f"oracle self-consistency on the variance: |a - b| max {np.max(np.abs(ref_var - var_b)):.3e}, "
      f"|a - c| max {np.max(np.abs(ref_var - var_c)):.3e}; device vs b {np.max(np.abs(var - var_b)):.3e}, "
      f"device vs c {np.max(np.abs(var - var_c)):.3e}", flush=True)
del W, Kc
eps = np.finfo(float).eps
em, ev = np.max(np.abs(mean - ref_mean)), np.max(np.abs(var - ref_var))
mt, vt = 1e-8 * np.max(np.abs(ref_mean)), 1e-8 * np.max(np.abs(ref_var))
print(f"PARITY N_tot={N} M={M}: mean rel err {em / np.max(np.abs(ref_mean)):.3e} (abs {em:.3e}, tol {mt:.3e}); "
      f"var rel err {ev / np.max(np.abs(ref_var)):.3e} (abs {ev:.3e}, tol {vt:.3e}); pass {bool(em <= mt and ev <= vt)}", flush=True)
