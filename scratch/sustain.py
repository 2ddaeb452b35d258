import sys; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np, time
from linpde_gp_amd import _engine
ctx = _engine.default_context()
rng = np.random.default_rng(0)
m = n = 8192; k = 512
A = rng.standard_normal((m, k)); B = rng.standard_normal((n, k)); C = np.zeros((m, n), order="F")
for reps in (1, 2, 5, 20, 100, 1, 100):
    _, ms = _engine.test_gemm(ctx, 0, 0, 0, -1.0, A, B, 1.0, C, k, reps=reps)
    print(f"reps={reps:4d}: {ms:.3f} ms/launch  {2.0*m*n*k/ms/1e9:.1f} TF")
    time.sleep(0.5)
