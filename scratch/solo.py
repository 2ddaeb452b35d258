import os, sys, time
from scratch_potrf import *   # noqa
from scratch_potrf import _engine
import numpy as np
wlx = wl
for solo, mst in ((0, 256), (0, 100000), (0, 256), (0, 100000)):
    ctx.set_option("solo_small", solo); ctx.set_option("min_supertiles", mst); ctx.set_option("nb", 512); ctx.set_option("lookahead", 1)
    ts = []
    for rep in range(3):
        mat = build(); t0 = time.perf_counter(); info = mat.potrf(); ctx.sync(); ts.append(time.perf_counter() - t0)
        if rep < 2: del mat
    # variance solve timing on this factor
    w = mat.solve_weights(np.ones(wlx.n_total))
    rhs = _engine.Rhs(ctx, mat, 4096)
    Xt = dev["test"]._lpgp_points
    for bi in range(5):
        k = covfuncs.DifferentiatedCovarianceFunction(prior.cov, *G._combine(prior.cov, coeffs[bi], {(0, 0): 1.0}))
        rhs.cross_assemble(k.lower(), pts[bi], Xt, bi)
    ctx.sync(); t0 = time.perf_counter(); rhs.trsm_lower(); ctx.sync(); tv = time.perf_counter() - t0
    print(f"min_supertiles={mst}: potrf {min(ts)*1e3:.2f} ms   variance trsm {tv*1e3:.2f} ms")
    del rhs, mat
