import sys; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
def run(m, n, k, tri):
    A = rng.standard_normal((m, k)); B = A if tri else rng.standard_normal((n, k))
    C = np.zeros((m, n), order="F")
    out = []
    for dense in (0, 1):
        ctx.set_option("dense_tiles", dense)
        _, ms = _hooks.test_gemm(ctx, 0, 0, tri, -1.0, A, B, 1.0, C, k, reps=5)
        fl = (m * (m + 1.0) * k) if tri else 2.0 * m * n * k
        out.append(fl / ms / 1e9)
    print(f"m={m} n={n} k={k} tri={tri}: legacy {out[0]:.1f} TF  dense {out[1]:.1f} TF", flush=True)
for n in (16384, 12288, 8192, 6144, 5120):
    run(n, n, 512, 1)
run(16384, 16384, 2048, 1)
run(16384, 4224, 512, 0)
run(12288, 4224, 512, 0)
run(8192, 4224, 512, 0)
run(16896 - 512, 512, 512, 0)
