import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
m, k = 12288, 512
A = rng.standard_normal((m, k)); C = np.zeros((m, m), order="F")
_, ms = _hooks.test_gemm(ctx, 0, 0, 1, -1.0, A, A, 1.0, C, k, reps=6)
print(f"LPGP_RESERVE_CUS={os.environ.get('LPGP_RESERVE_CUS')} stream={os.environ.get('LPGP_TEST_GEMM_STREAM')}: {ms:.3f} ms {m*(m+1.0)*k/ms/1e9:.1f} TF", flush=True)
