"""Worker of tests/test_host_asan.py: runs inside a python started with LD_PRELOAD=libasan.so, loads the
AddressSanitizer build of the host-only lowering code (`build.sh --host-asan`) and drives it over the
descriptor zoo: every descriptor the host mirror produces for the kernels / operators of the parity
tests (1-D Matern all (nu, a, b), ExpQuad, 2-D Poisson, heat, 3-D sum kernel, isotropic Matern with
directional derivatives, Robin-type sums) + malformed descriptors.  The lowered descriptor is evaluated
on the CPU by the very evaluation core the GPU kernels use (csrc/eval_entries.h) and compared with the
oracle; ASan / UBSan abort the process on any memory error in the lowering."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))

from linpde_gp_amd import _lib  # noqa: E402  (KDesc layout + the host mirror's lowering)
from linpde_gp_amd.linfuncops import diffops  # noqa: E402
from linpde_gp_amd.randprocs import covfuncs as cf  # noqa: E402
from oracle import covfuncs as ocf  # noqa: E402

lib = C.CDLL(sys.argv[1])
pd = C.POINTER(C.c_double)
lib.lpgp_host_kernel_matrix.restype = C.c_int
lib.lpgp_host_kernel_matrix.argtypes = [C.POINTER(_lib.KDesc), C.c_int32, pd, C.c_int64, pd, C.c_int64, pd]
lib.lpgp_host_kernel_matrix_fact.restype = C.c_int
lib.lpgp_host_kernel_matrix_fact.argtypes = lib.lpgp_host_kernel_matrix.argtypes
lib.lpgp_host_kernel_diag.restype = C.c_int
lib.lpgp_host_kernel_diag.argtypes = [C.POINTER(_lib.KDesc), C.c_int32, pd]
lib.lpgp_host_last_error.restype = C.c_char_p


def host_matrix(k, X0, X1):
    arr = _lib.make_kdesc_array(k.lower())
    X0 = np.ascontiguousarray(X0, dtype=np.double).reshape(len(X0), -1)
    X1 = np.ascontiguousarray(X1, dtype=np.double).reshape(len(X1), -1)
    out = np.full((X0.shape[0], X1.shape[0]), np.nan)
    rc = lib.lpgp_host_kernel_matrix(arr, len(arr), _lib.as_pd(X0), X0.shape[0], _lib.as_pd(X1), X1.shape[0], _lib.as_pd(out))
    assert rc == 0, lib.lpgp_host_last_error()
    # the factored evaluation (per-point exponentials with double-double arguments; what the device kernels run on tiles of
    # moderate extent) must agree with the per-entry one to a few ulps of the block maximum, whatever |a (x - x0)| is
    out_f = np.full_like(out, np.nan)
    rc = lib.lpgp_host_kernel_matrix_fact(arr, len(arr), _lib.as_pd(X0), X0.shape[0], _lib.as_pd(X1), X1.shape[0], _lib.as_pd(out_f))
    assert rc == 0, lib.lpgp_host_last_error()
    scale = max(np.max(np.abs(out)), 1e-300)
    assert np.max(np.abs(out_f - out)) <= 2e-14 * scale, ("factored vs per-entry evaluation", np.max(np.abs(out_f - out)) / scale)
    global worst_fact
    worst_fact = max(worst_fact, np.max(np.abs(out_f - out)) / scale)
    v = C.c_double()
    assert lib.lpgp_host_kernel_diag(arr, len(arr), C.byref(v)) == 0
    return out, v.value


def check_exp_neg():
    """`lpgp_exp_neg` (table + degree-4 polynomial, csrc/eval_entries.h) against 40-digit arithmetic: below 0.53 units in the
    last place over the whole range the kernels use it on, exact at 0, 0 beyond the clamp, monotone across table steps."""
    import mpmath as mp
    mp.mp.dps = 40
    lib.lpgp_host_exp_neg.restype = None
    lib.lpgp_host_exp_neg.argtypes = [pd, C.c_int64, pd]
    rng = np.random.default_rng(5)
    s = np.concatenate([rng.uniform(0, 1e-3, 3000), rng.uniform(0, 2, 6000), rng.uniform(0, 40, 6000), rng.uniform(0, 700, 3000),
                        np.arange(0, 64) * (np.log(2) / 256), [0.0, 5e-324, 1e-300, 708.0, 744.0, 800.0, 1e10, 1e300, np.inf]])
    out = np.empty_like(s)
    lib.lpgp_host_exp_neg(_lib.as_pd(s), len(s), _lib.as_pd(out))
    worst = 0.0
    for si, oi in zip(s, out):
        ex = mp.exp(-mp.mpf(float(si))) if np.isfinite(si) else mp.mpf(0)
        exd = float(ex)
        if exd < 2.3e-308:
            assert abs(oi - exd) <= 5e-324 * 2 or oi <= 2.3e-308, (si, oi, exd)
            continue
        ulp = np.spacing(exd)
        worst = max(worst, float(abs(mp.mpf(float(oi)) - ex) / ulp))
    assert worst <= 0.53, worst
    assert out[s == 0.0][0] == 1.0 and out[-1] == 0.0 and out[-2] == 0.0
    grid = np.linspace(0.0, 30.0, 200001)
    og = np.empty_like(grid)
    lib.lpgp_host_exp_neg(_lib.as_pd(grid), len(grid), _lib.as_pd(og))
    assert np.all(np.diff(og) <= 0.0)
    return worst


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


checked = 0
worst_fact = 0.0


def check(k, okern, L0, L1, X0, X1, tol=1e-12):
    global checked
    got, diag = host_matrix(k, X0, X1)
    ref = ocf.LkL(okern, L0, L1, X0.reshape(len(X0), -1), X1.reshape(len(X1), -1))
    assert rel(got, ref) < tol, (rel(got, ref), L0, L1)
    Xd = X0.reshape(len(X0), -1)[:3]
    assert abs(diag - ocf.k_diag(okern, L0, L1, Xd)[0]) <= 1e-12 * max(abs(diag), 1.0)
    checked += 1


rng = np.random.default_rng(20261002)
# ---- 1-D Matern: every (nu, a, b) with a + b <= 2p, a, b <= 2p (cases of test_matern_1d_blocks) ----
for nu in (0.5, 1.5, 2.5, 3.5, 4.5):
    p = int(nu - 0.5)
    k = cf.Matern((), nu=nu, lengthscales=0.9)
    X0, X1 = rng.uniform(-3, 3, (37, 1)), rng.uniform(-3, 3, (21, 1))
    for a in range(0, 2 * p + 1):
        for b in range(0, 2 * p + 1 - a):
            if a > 4 or b > 4:
                continue
            kk = diffops.Derivative(a)(diffops.Derivative(b)(k, argnum=1), argnum=0)
            check(kk, [(1.0, [("matern", nu, 0.9)])], {(a,): 1.0}, {(b,): 1.0}, X0, X1)
# ---- ExpQuad, all 9 order pairs ----
k = 4.0 * cf.ExpQuad((), lengthscales=0.25)
X0, X1 = rng.uniform(-1, 1, (33, 1)), rng.uniform(-1, 1, (17, 1))
for a in range(3):
    for b in range(3):
        kk = diffops.Derivative(a)(diffops.Derivative(b)(k, argnum=1), argnum=0)
        check(kk, [(4.0, [("expquad", 0.25)])], {(a,): 1.0}, {(b,): 1.0}, X0, X1)
# ---- Poisson 2-D and heat ----
X0, X1 = rng.uniform(-1, 1, (40, 2)), rng.uniform(-1, 1, (23, 2))
k = 4.0 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=0.7))
okern = [(4.0, [("matern", 2.5, 1.0), ("matern", 2.5, 0.7)])]
D = -1.0 * diffops.Laplacian((2,))
lap, ident = {(2, 0): -1.0, (0, 2): -1.0}, ocf.identity(2)
check(k, okern, ident, ident, X0, X1)
check(D(k, argnum=1), okern, ident, lap, X0, X1)
check(D(k, argnum=0), okern, lap, ident, X0, X1)
check(D(D(k, argnum=1), argnum=0), okern, lap, lap, X0, X1)
kh = cf.TensorProduct(cf.Matern((), nu=1.5, lengthscales=2.5), cf.Matern((), nu=2.5, lengthscales=2.0))
okh = [(1.0, [("matern", 1.5, 2.5), ("matern", 2.5, 2.0)])]
H = diffops.HeatOperator((2,), alpha=0.1)
heat = {(1, 0): 1.0, (0, 2): -0.1}
check(H(H(kh, argnum=1), argnum=0), okh, heat, heat, X0, X1)
check(H(kh, argnum=0), okh, heat, ident, X0, X1)
dd = diffops.DirectionalDerivative([0.3, -1.2])
check(dd(H(kh, argnum=1), argnum=0), okh, {(1, 0): 0.3, (0, 1): -1.2}, heat, X0, X1)
# ---- 3-D sum of an ExpQuad and a product Matern under the Laplacian (two groups, 9 terms each) ----
X0, X1 = rng.normal(size=(25, 3)), rng.normal(size=(19, 3))
k3 = 1.5 * cf.ExpQuad((3,), lengthscales=[0.4, 1.3, 0.9]) + 0.5 * cf.TensorProduct(
    *(cf.Matern((), nu=2.5, lengthscales=l) for l in (1.0, 2.0, 0.5)))
ok3 = [(1.5, [("expquad", 0.4), ("expquad", 1.3), ("expquad", 0.9)]),
       (0.5, [("matern", 2.5, 1.0), ("matern", 2.5, 2.0), ("matern", 2.5, 0.5)])]
L3 = diffops.Laplacian((3,))
lap3 = {(2, 0, 0): 1.0, (0, 2, 0): 1.0, (0, 0, 2): 1.0}
check(L3(L3(k3, argnum=1), argnum=0), ok3, lap3, lap3, X0, X1)
check(L3(k3, argnum=0), ok3, lap3, ocf.identity(3), X0, X1)
# ---- 4-D product kernel (the largest parity-class table: 16 classes) ----
X0, X1 = rng.normal(size=(12, 4)), rng.normal(size=(9, 4))
k4 = cf.TensorProduct(*(cf.Matern((), nu=1.5, lengthscales=l) for l in (1.0, 2.0, 0.5, 1.5)))
ok4 = [(1.0, [("matern", 1.5, l) for l in (1.0, 2.0, 0.5, 1.5)])]
d4 = diffops.DirectionalDerivative([0.5, -1.0, 2.0, 0.25])
dd4 = {(1, 0, 0, 0): 0.5, (0, 1, 0, 0): -1.0, (0, 0, 1, 0): 2.0, (0, 0, 0, 1): 0.25}
check(d4(d4(k4, argnum=1), argnum=0), ok4, dd4, dd4, X0, X1)
# ---- isotropic Matern, identity / directional derivatives / Robin-type sums ----
X0, X1 = rng.uniform(-3, 3, (30, 3)), rng.uniform(-3, 3, (22, 3))
ls = np.array([0.7, 1.0, 2.0])
for nu in (1.5, 2.5, 3.5, 4.5):
    ki = cf.Matern((3,), nu=nu, lengthscales=ls)
    oki = [(1.0, [("matern_iso", nu, ls)])]
    v0, v1 = rng.standard_normal(3), rng.standard_normal(3)
    c0 = {tuple(int(i == j) for i in range(3)): float(v0[j]) for j in range(3)}
    c1 = {tuple(int(i == j) for i in range(3)): float(v1[j]) for j in range(3)}
    check(ki, oki, ocf.identity(3), ocf.identity(3), X0, X1, 1e-11)
    check(diffops.DirectionalDerivative(v1)(ki, argnum=1), oki, ocf.identity(3), c1, X0, X1, 1e-11)
    check(diffops.DirectionalDerivative(v0)(ki, argnum=0), oki, c0, ocf.identity(3), X0, X1, 1e-11)
    if nu > 1.5:
        check(diffops.DirectionalDerivative(v0)(diffops.DirectionalDerivative(v1)(ki, argnum=1), argnum=0),
              oki, c0, c1, X0, X1, 1e-11)

# ---- malformed descriptors: rejected with a message, nothing read or written out of bounds ----
good = _lib.make_kdesc_array(k.lower())
out = np.zeros((2, 2))
Xs = np.zeros((2, 2))


def expect_reject(mutate, ngroups=1):
    arr = _lib.make_kdesc_array(k.lower())
    mutate(arr[0])
    rc = lib.lpgp_host_kernel_matrix(arr, ngroups, _lib.as_pd(Xs), 2, _lib.as_pd(Xs), 2, _lib.as_pd(out))
    assert rc != 0 and lib.lpgp_host_last_error(), "malformed descriptor accepted"


expect_reject(lambda kd: setattr(kd, "d", 0))
expect_reject(lambda kd: setattr(kd, "d", 5))
expect_reject(lambda kd: setattr(kd, "nterms", 0))
expect_reject(lambda kd: setattr(kd, "nterms", _lib.MAXT + 1))
expect_reject(lambda kd: kd.p.__setitem__(0, 7))
expect_reject(lambda kd: kd.p.__setitem__(1, -1))
expect_reject(lambda kd: kd.family.__setitem__(1, 9))
expect_reject(lambda kd: kd.lengthscale.__setitem__(0, 0.0))
expect_reject(lambda kd: kd.terms[0].n0.__setitem__(0, -1))
expect_reject(lambda kd: kd.terms[0].n1.__setitem__(1, 13))
expect_reject(lambda kd: None, ngroups=0)
expect_reject(lambda kd: None, ngroups=_lib.MAXG + 1)


def iso_order2(kd):
    kd.family[0] = kd.family[1] = _lib.MATERN_ISO
    kd.terms[0].n0[0] = 2


expect_reject(iso_order2)


def coef_overflow(kd):       # 4-D, degree 6 per dimension: 7^4 coefficients per parity class, four classes > the table
    kd.d = 4
    for j in range(4):
        kd.family[j], kd.p[j], kd.lengthscale[j] = _lib.MATERN_HALFINT, 6, 1.0
    kd.nterms = 4
    for t in range(4):
        kd.terms[t].coef = 1.0
        for j in range(4):
            kd.terms[t].n0[j], kd.terms[t].n1[j] = int(j == t), 0


expect_reject(coef_overflow)
worst_exp = check_exp_neg()
print(f"host-asan worker: lpgp_exp_neg within {worst_exp:.3f} ulp of exp(-s)")
print(f"host-asan worker: {checked} descriptors evaluated against the oracle, 14 malformed ones rejected; "
      f"factored vs per-entry evaluation: worst {worst_fact:.2e} of the block maximum")
