"""Column tiles per workgroup of assemble_fast_kernel (`asm_ct`): rate on the 4096 x 16384 cross-covariance of c3 and on a
symmetric 8192 x 8192 diagonal block (lower triangle), bit-identity with one tile per workgroup."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems
ctx = _engine.default_context()
wl = problems.poisson_2d()
k = problems.build_prior(wl).cov
D = problems.operator_of(wl.observations[-1].op, 2)
Xobs = np.ascontiguousarray(wl.observations[-1].X)
Xt = wl.Xtest
kk = D(k, argnum=1)
ref = None
for ct in (1, 2, 4, 8, 16, 1, 4):
    ctx.set_option("asm_ct", ct)
    best = 1e9
    for rep in range(5):
        ctx.profile_reset(); ctx.profile_enable(["assemble"])
        M = kk.matrix(Xt, Xobs)
        ctx.sync(); p = ctx.profile_get()["assemble"]; ctx.profile_enable(False)
        best = min(best, p["ms"])
    if ref is None:
        ref = M
    print(f"asm_ct {ct:2d}: cross-covariance 4096 x 16384 best of 5 {best:.4f} ms -> {p['bytes'] / best / 1e6:.0f} GB/s; identical to ct=1: {np.array_equal(M, ref)}", flush=True)
kd = D(D(k, argnum=1), argnum=0)
X = Xobs[:8192 - 37]
refd = None
for ct in (1, 4):
    ctx.set_option("asm_ct", ct)
    Md = kd.matrix(X, X)
    refd = Md if refd is None else refd
    print(f"asm_ct {ct}: ragged 8155 x 8155 block identical: {np.array_equal(Md, refd)}")
ctx.set_option("asm_ct", 4)
