# one-shot potrf of the c3 Gram matrix (for rocprofv3 --kernel-trace); prints wall time
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems
from linpde_gp_amd.randprocs import covfuncs
import linpde_gp_amd.randprocs._gaussian_process as G
ctx = _engine.default_context()
for kv in os.environ.get("LPGP_OPTS", "").split(","):
    if "=" in kv:
        k_, v_ = kv.split("="); ctx.set_option(k_, int(v_)); print("option", k_, v_)
wl = problems.poisson_2d(128, m_side=64)
dev = problems.upload(wl)
prior = problems.build_prior(wl)
pts = [d_._lpgp_points for d_ in dev["obs"]]
coeffs = [o.op for o in wl.observations]
def build():
    mat = _engine.GramMatrix(ctx, wl.n_total)
    for bi in range(5):
        mat.add_block(pts[bi].n)
        for bj in range(bi + 1):
            k = covfuncs.DifferentiatedCovarianceFunction(prior.cov, *G._combine(prior.cov, coeffs[bi], coeffs[bj]))
            mat.assemble(k.lower(), pts[bi], None if bi == bj else pts[bj], bi, bj)
        if wl.observations[bi].noise_var: mat.add_diag(bi, None, wl.observations[bi].noise_var)
    ctx.sync()
    return mat
for rep in range(3):
    mat = build(); t0 = time.perf_counter(); info = mat.potrf(); ctx.sync(); t = time.perf_counter() - t0
    print(f"potrf {t*1e3:.2f} ms info={info}")
    if rep < 2: del mat
K = _engine.Rhs(ctx, mat, 4096) if hasattr(_engine, "Rhs") else None
