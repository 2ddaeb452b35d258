# wall time of the blocked Cholesky on small matrices (entirely bound by the panel chain)
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine
from linpde_gp_amd.randprocs import covfuncs as cf
ctx = _engine.default_context()
k = cf.Matern((), nu=2.5, lengthscales=0.3)
for n in (1024, 2048, 4096, 8192):
    X = np.linspace(-1, 1, n)[:, None]
    P = _engine.Points(ctx, X)
    best = 1e9
    for rep in range(4):
        mat = _engine.GramMatrix(ctx, n)
        mat.add_block(n)
        mat.assemble(k.lower(), P, None, 0, 0)
        mat.add_diag(0, None, 1e-3)
        ctx.sync(); t0 = time.perf_counter(); info = mat.potrf(); ctx.sync()
        best = min(best, time.perf_counter() - t0)
        del mat
    T = n // 128
    print(f"n={n} tiles={T} potrf {best*1e3:.3f} ms  per tile {best*1e6/T:.1f} us  info {info}  ({n**3/3/best/1e12:.1f} TFLOP/s)")
