"""Steady-state rate of the forward substitution's update product C(m x 4096) -= A(m x K) B(K x 4096) (A not transposed, B
transposed storage: the kernel symbol gemm_f64_kernel<false,true,0> / gemm3_f64_kernel<true,0>), 200+ back-to-back launches."""
import os, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
n = 4096
out = []
for m, k in ((15872, 512), (8192, 512), (4096, 512), (15872, 1024)):
    A = rng.standard_normal((m, k)); B = rng.standard_normal((k, n)); C = np.zeros((m, n), order="F")
    _, ms = _hooks.test_gemm(ctx, 0, 1, 0, -1.0, A, B, 1.0, C, k, reps=max(100, int(300 * 15872 * 512 / (m * k))))
    out.append(f"{m} x {n} x {k}: {ms:.3f} ms = {2.0 * m * n * k / ms / 1e9:.1f}")
print(f"LPGP_GEMM3={os.environ.get('LPGP_GEMM3', '768 (default)')}: TFLOP/s  " + ";  ".join(out), flush=True)
