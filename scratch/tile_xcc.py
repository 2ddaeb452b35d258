import sys, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems
from linpde_gp_amd._lib import lib, check
ctx = _engine.default_context()
wl = problems.poisson_2d(128, m_side=64)
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl); prior = problems.build_prior(wl)
out = (C.c_int32 * 8)()
check(lib.lpgp_debug_tile_xcc(ctx._h, out, 1))
for rep in range(3):
    problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
    check(lib.lpgp_debug_tile_xcc(ctx._h, out, 1))
    print("tile Cholesky workgroups per XCD:", list(out))
