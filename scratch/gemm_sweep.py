import sys; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
def run(m, n, k, tri, ta=0, tb=0):
    A = rng.standard_normal((k, m) if ta else (m, k)); B = rng.standard_normal((k, n) if tb else (n, k))
    C = np.zeros((m, n), order="F")
    _, ms = _hooks.test_gemm(ctx, ta, tb, tri, -1.0, A, B, 1.0, C, k, reps=5)
    fl = (m * (m + 1.0) * k) if tri else 2.0 * m * n * k
    print(f"m={m} n={n} k={k} tri={tri} ta={ta} tb={tb}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF")
run(2048, 2048, 8192, 0)      # 256 tiles: one per CU, long K
run(4096, 4096, 4096, 0)      # 1024 tiles, long K
run(4096, 2048, 512, 0)       # 512 tiles = exactly 2 per CU, short K
run(8192, 8192, 512, 0)
run(8192, 8192, 512, 1)
run(16384, 16384, 512, 1)
run(16384, 4096, 512, 0, 0, 1)
run(8192, 8192, 128, 1)
