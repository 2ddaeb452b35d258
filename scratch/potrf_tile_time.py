"""potrf_tile_kernel alone on an idle chip: HIP-event time over 50 launches (profiling slot "potrf_tile"), accuracy against LAPACK."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np, scipy.linalg
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
M = rng.standard_normal((128, 160)); A = M @ M.T + 1e-3 * np.eye(128)
ctx.profile_reset(); ctx.profile_enable(["potrf_tile"])
for _ in range(50):
    L, Linv, info = _hooks.test_potrf_tile(ctx, A)
ctx.sync(); p = ctx.profile_get()["potrf_tile"]; ctx.profile_enable(False)
Lref = np.linalg.cholesky(A)
eL = np.max(np.abs(np.tril(L) - Lref)) / np.max(np.abs(Lref))
eI = np.max(np.abs(Linv @ Lref - np.eye(128)))
# backward error of the factor and of the explicit inverse
eb = np.max(np.abs(np.tril(L) @ np.tril(L).T - A)) / np.max(np.abs(A))
ebref = np.max(np.abs(Lref @ Lref.T - A)) / np.max(np.abs(A))
print(f"potrf_tile: {p['ms'] / p['launches'] * 1e3:.1f} us per launch ({p['launches']} launches); |L - L_lapack| {eL:.2e}, |Linv L - I| {eI:.2e}, backward error {eb:.2e} (LAPACK {ebref:.2e})")
