import sys; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
ctx = _engine.default_context()
rng = np.random.default_rng(0)
for n in (16384, 12288, 8192):
    k = 512
    P = rng.standard_normal((n, k)); C = np.zeros((n, n), order="F")
    _, ms = _engine.test_gemm(ctx, 0, 0, 1, -1.0, P, P, 1.0, C, k, reps=8)
    print(f"SYRK n={n}: {ms:.3f} ms {n*(n+1.0)*k/ms/1e9:.1f} TF")
