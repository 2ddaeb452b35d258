"""A/B of the three-workgroups-per-CU GEMM (gemm3_f64_kernel, option `gemm3`) against the two-resident one:
stand-alone SYRK / GEMM shapes on random data, then nothing else (the in-situ A/B is bench.py under LPGP_GEMM3=0/1)."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
ctx.set_option("small_tiles_max", 0)
ctx.set_option("gemm3_fact", 1)
def run(m, n, k, tri, tb=0):
    A = rng.standard_normal((m, k)); B = A if tri else (rng.standard_normal((k, n)) if tb else rng.standard_normal((n, k)))
    C = np.zeros((m, n), order="F")
    out = []
    for g3 in (0, 1, 0, 1):
        ctx.set_option("gemm3", g3)
        _, ms = _hooks.test_gemm(ctx, 0, tb, tri, -1.0, A, B, 1.0, C, k, reps=8)
        fl = (m * (m + 1.0) * k) if tri else 2.0 * m * n * k
        out.append(fl / ms / 1e9)
    print(f"m={m} n={n} k={k} tri={tri} tb={tb}: two-resident {out[0]:.1f} / {out[2]:.1f} TF   three-resident {out[1]:.1f} / {out[3]:.1f} TF", flush=True)
# correctness first
P = rng.standard_normal((640, 512)); C0 = rng.standard_normal((640, 640))
ctx.set_option("gemm3", 1)
o, _ = _hooks.test_gemm(ctx, 0, 0, 1, -1.0, P, P, 1.0, C0, 512)
t = np.arange(640) // 128
low = t[:, None] >= t[None, :]
print("gemm3 syrk max err", np.max(np.abs((o - (C0 - P @ P.T))[low])), "upper untouched", np.array_equal(o[~low], C0[~low]))
Bm = rng.standard_normal((384, 80)); Am = rng.standard_normal((256, 80)); C1 = rng.standard_normal((256, 384))
o, _ = _hooks.test_gemm(ctx, 0, 0, 0, -1.5, Am, Bm, 0.5, C1, 80)
print("gemm3 gemm k=80 max err", np.max(np.abs(o - (0.5 * C1 - 1.5 * Am @ Bm.T))))
Bk = rng.standard_normal((80, 384))
o, _ = _hooks.test_gemm(ctx, 0, 1, 0, -1.5, Am, Bk, 0.5, C1, 80)
print("gemm3 gemm tb=1 k=80 max err", np.max(np.abs(o - (0.5 * C1 - 1.5 * Am @ Bk))))
run(16384, 4224, 512, 0, 1)
run(8192, 4224, 512, 0, 1)
run(16384, 4224, 1024, 0, 1)
run(16384, 16384, 512, 1)
run(12288, 12288, 512, 1)
run(8192, 8192, 512, 1)
run(4096, 4096, 512, 1)
run(16384, 16384, 1024, 1)
run(16384, 16384, 2048, 1)
run(8192, 8192, 512, 0)
run(16384, 4096, 512, 0)
