# distributed trailing update of ONE rank: a launch per owned panel against one ownership-filtered launch
import os, sys; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
ctx = _engine.default_context()
rng = np.random.default_rng(0)
k, w = 512, 4
for T, P in ((150, 2), (150, 4)):      # (the NumPy reference of larger cases takes minutes on the host)
    n = T * 128
    A = rng.standard_normal((n, k))
    C0 = np.zeros((n, n), order="F")
    rank = 1 % P
    # (i) one launch per owned panel: rows [q0, T) x columns [q0, q0 + w)
    os.environ.pop("LPGP_TEST_OWN", None)
    ms_sep = 0.0
    nl = 0
    for q0 in range(0, T, w):
        if (q0 // w) % P != rank: continue
        q1 = min(q0 + w, T)
        Cs = np.zeros(((T - q0) * 128, (q1 - q0) * 128), order="F")
        _, ms = _engine.test_gemm(ctx, 0, 0, 1, -1.0, A[q0 * 128:], A[q0 * 128:q1 * 128], 1.0, Cs, k, reps=3)
        ms_sep += ms; nl += 1
    # (ii) one filtered launch
    os.environ["LPGP_TEST_OWN"] = f"{P},{rank},0,{w}"
    Cm, ms_m = _engine.test_gemm(ctx, 0, 0, 1, -1.0, A, A, 1.0, C0, k, reps=3)
    os.environ.pop("LPGP_TEST_OWN", None)
    # correctness of the filtered launch: owned columns updated (lower part), the others untouched
    ref = -(A @ A.T)
    ok = True
    for c in range(T):
        blk = np.tril(Cm)[:, c * 128:(c + 1) * 128]
        want = np.tril(ref)[:, c * 128:(c + 1) * 128] if (c // w) % P == rank else 0.0
        ok &= np.allclose(blk, want, rtol=0, atol=1e-9)
    print(f"T={T} P={P}: {nl} launches {ms_sep:.3f} ms   one filtered launch {ms_m:.3f} ms   correct={ok}", flush=True)
