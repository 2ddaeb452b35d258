"""Where does the GPU-vs-oracle difference of the posterior at full c3 size come from?
(a) Gram entries (assembly) or (b) factorisation / solves.  Runs on the GPU box.
  G_gpu: the matrix the device assembled (captured right before lpgp_potrf), G_cpu: the oracle's.
  For each of them: LAPACK solve in fp64 and a refined ("exact for that matrix") solve with long-double
  residuals; compared with the device posterior and with each other."""
import sys, os, time
import numpy as np, scipy.linalg
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
import linpde_gp_amd as lp
from linpde_gp_amd import problems, _engine
from oracle import workloads as owl, gp as ogp, covfuncs as ocf

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
wl = {"c3": lambda: problems.poisson_2d(), "c3s": lambda: problems.poisson_2d(64, m_side=32),
      "c5m": lambda: problems.heat_1d(nt=128, nx=64, m_side=32)}[which]()
lp.config.gram_capacity_hint = wl.n_total
captured = {}
orig = _engine.GramMatrix.potrf
def potrf(self):            # all blocks are assembled first (nothing factored), the Gram is copied out, then ONE factorisation
    if len(self.block_sizes) < len(wl.observations):
        return 0
    captured["G"] = self.todense("gram")
    return orig(self)
u_app, mean_app, var_app = problems.condition_and_predict(wl)      # the product path: block append
_engine.GramMatrix.potrf = potrf
u, mean, var = problems.condition_and_predict(wl)
_engine.GramMatrix.potrf = orig
Gg = captured["G"]
print(f"device: block-append path vs one-shot factorisation: mean {np.max(np.abs(mean_app - mean)) / np.max(np.abs(mean)):.2e}, "
      f"var {np.max(np.abs(var_app - var)) / np.max(np.abs(var)):.2e}")
blocks = owl.blocks_of(wl)
t = time.time(); Gc = ogp.gram(wl.kernel, blocks); print("oracle gram", time.time() - t, "s")
r = ogp.residual(blocks)
Kc = ogp.cross_cov(wl.kernel, blocks, wl.Xtest)
off = np.cumsum([0] + [b.n for b in blocks])
print("Gram entry differences, relative to the block maximum:")
for i in range(len(blocks)):
    for j in range(i + 1):
        a, b = Gg[off[i]:off[i+1], off[j]:off[j+1]], Gc[off[i]:off[i+1], off[j]:off[j+1]]
        print(f"  block ({i},{j}) {a.shape}: {np.max(np.abs(a - b)) / np.max(np.abs(b)):.2e}")
# cross-covariance as the device evaluates it (per-entry kernel)
prior = problems.build_prior(wl)

def solve_refined(G, rhs, iters=4):
    c = scipy.linalg.cholesky(G, lower=True, check_finite=False)
    x = scipy.linalg.cho_solve((c, True), rhs, check_finite=False)
    x0 = x.copy()
    Gl = G.astype(np.longdouble)
    for it in range(iters):
        res = (rhs.astype(np.longdouble) - Gl @ x.astype(np.longdouble)) if x.ndim == 1 else None
        dx = scipy.linalg.cho_solve((c, True), np.asarray(res, dtype=np.double), check_finite=False)
        x = (x.astype(np.longdouble) + dx).astype(np.double) if False else x + dx
        print(f"    refinement {it}: |dx|/|x| = {np.linalg.norm(dx) / np.linalg.norm(x):.2e}")
    return x0, x, c

rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
print("fp64 LAPACK vs refined solve (long-double residuals):")
w_c0, w_c, chol_c = solve_refined(Gc, r)
w_g0, w_g, chol_g = solve_refined(Gg, r)
m_c0, m_c, m_g0, m_g = Kc @ w_c0, Kc @ w_c, Kc @ w_g0, Kc @ w_g
print("posterior MEAN, max abs difference / max|mean|:")
print(f"  device               vs oracle(LAPACK, G_cpu)      {rel(mean, m_c0):.2e}")
print(f"  device               vs refined(G_cpu)             {rel(mean, m_c):.2e}")
print(f"  device               vs refined(G_gpu)             {rel(mean, m_g):.2e}")
print(f"  device               vs LAPACK(G_gpu)              {rel(mean, m_g0):.2e}")
print(f"  oracle(LAPACK,G_cpu) vs refined(G_cpu)             {rel(m_c0, m_c):.2e}")
print(f"  LAPACK(G_gpu)        vs refined(G_gpu)             {rel(m_g0, m_g):.2e}")
print(f"  refined(G_gpu)       vs refined(G_cpu)             {rel(m_g, m_c):.2e}   <- effect of the Gram-entry differences alone")
# variance on a few points: fp64 LAPACK on both matrices
idx = np.linspace(0, wl.Xtest.shape[0] - 1, 64).astype(int)
kxx = float(sum(sc for sc, _ in wl.kernel))
Vc = scipy.linalg.solve_triangular(chol_c, Kc[idx].T, lower=True, check_finite=False)
Vg = scipy.linalg.solve_triangular(chol_g, Kc[idx].T, lower=True, check_finite=False)
vc, vg = kxx - np.sum(Vc * Vc, 0), kxx - np.sum(Vg * Vg, 0)
print("posterior VARIANCE on 64 points, max abs difference / max|var| (max var %.3e):" % np.max(vc))
print(f"  device vs LAPACK(G_cpu) {rel(var[idx], vc):.2e};  device vs LAPACK(G_gpu) {rel(var[idx], vg):.2e};  LAPACK(G_gpu) vs LAPACK(G_cpu) {rel(vg, vc):.2e}")
print("cond estimate: max diag(L)/min diag(L) squared = %.2e" % ((np.max(np.diag(chol_c)) / np.min(np.diag(chol_c))) ** 2))
