"""Persistent form of the rank-512 trailing update (LPGP_GEMM_PERSIST=1) on the masked / unmasked stream: rate, and the result
against the ordinary form's (bit for bit: the tiles are the same, only who computes them changes)."""
import os, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
out = []
for m in (16384, 12288, 8192):
    A = rng.standard_normal((m, 512)); C = np.zeros((m, m), order="F")
    best = 1e9
    for rep in range(3):
        R, ms = _hooks.test_gemm(ctx, 0, 0, 1, -1.0, A, A, 1.0, C, 512, reps=5)
        best = min(best, ms)
    ref = -(A[:2048] @ A[:2048].T)
    err = np.max(np.abs(np.tril(R[:2048, :2048]) - np.tril(ref)))
    err2 = np.max(np.abs(np.tril(R[-1024:, -1024:]) + np.tril(A[-1024:] @ A[-1024:].T)))
    out.append(f"{m}: {m * (m + 1.0) * 512 / best / 1e9:.1f} (err {max(err, err2):.1e})")
print(f"persist {os.environ.get('LPGP_GEMM_PERSIST', '0')} reserve {os.environ.get('LPGP_RESERVE_CUS', '8'):>3s} stream {os.environ.get('LPGP_TEST_GEMM_STREAM', '0')}: TFLOP/s  " + "   ".join(out), flush=True)
