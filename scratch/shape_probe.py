import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
ctx = _engine.default_context()
rng = np.random.default_rng(0)
def run(m, n, k, tri, tb=0):
    A = rng.standard_normal((m, k)); 
    B = A if tri else (rng.standard_normal((k, n)) if tb else rng.standard_normal((n, k)))
    C = np.zeros((m, n), order="F")
    _, ms = _engine.test_gemm(ctx, 0, tb, tri, -1.0, A, B, 1.0, C, k, reps=5)
    fl = (m * (m + 1.0) * k) if tri else 2.0 * m * n * k
    print(f"m={m} n={n} k={k} tri={tri} tb={tb}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF ({fl/ms/1e9/78.6*100:.0f} %)", flush=True)
for k in (512, 768, 1024):
    run(12288, 12288, k, 1)
for k in (512, 768, 1024):
    run(12288, 4224, k, 0, 1)
for k in (512, 768):
    run(12288, 6144, k, 0, 0)
