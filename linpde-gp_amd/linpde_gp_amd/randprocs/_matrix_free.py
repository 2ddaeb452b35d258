"""GP posteriors WITHOUT a factor: every product with the Gram matrix re-evaluates its entries on the GPU
(`lpgp_kernel_matvec`), solves are preconditioned conjugate gradients on top of that product.

This is the slot the reference fills with KeOps lazy tensors (`_keops_lazy_tensor`,
`covfuncs/linfuncops/diffops/_matern.py:112-135,231-264`; probnum's `LinearOperator.solve` then iterates on the lazy product):
observation sets beyond the dense memory of one device -- 190 000 points are a 289-GB Gram matrix -- and the CPU-die
experiment's 2-D model (`experiments/0001_cpu_stationary_2d.ipynb`).  Same surface as `ConditionalGaussianProcess`:
`mean`, `var`, `std`, `predict`, `cov.matrix`, `representer_weights`, `gram` (`@`, `solve`), re-entrant
`condition_on_observations` (the previous weights warm-start the next solve).

What runs where: every kernel entry -- Gram products, cross-covariance products, the rows the preconditioner is built from --
is evaluated by the HIP kernels behind `lpgp_kernel_matvec(_dev)` / `lpgp_kernel_matrix`.  Since round 6 the ITERATION is
device-resident too (`pcg_device`): iterates, residuals and search directions live in HBM (`lpgp_dvec`), a product is launches
only (`lpgp_kernel_matvec_dev` per block pair + the noise diagonals), and the vector algebra -- column dots, step lengths, the
low-rank preconditioner -- is `lpgp_pcg_step`; the host reads back the relative residuals (8 bytes per column and iteration).
The reference's KeOps operands stay on the device in the same way (`diffops/_matern.py:112-135`).  The host loop `pcg` is kept
for Gram matrices with DENSE noise blocks and as the statement of the algorithm the device runs.  There is no dense N x N
object anywhere.

Preconditioner: rank-r pivoted Cholesky `G ~ L^T L` (r rows of G, greedy on the remaining diagonal) plus `delta I`, applied by
the Woodbury identity -- the standard choice for kernel matrices with a noise floor; `delta` is the mean of the diagonal the
low-rank part leaves unexplained.
"""

from __future__ import annotations

import numpy as np

from .. import _engine, config, functions, randvars
from . import covfuncs


class GramProduct:
    """`G = [ (L_i k L_j'^*)(X_i, X_j) ]_ij + blockdiag(noise)` as products (`_conditional.py:357-394` without the matrix)."""

    def __init__(self, ctx, base_cov, blocks):
        from ._gaussian_process import _lowered, _noise_diag
        self.ctx, self.base, self.blocks = ctx, base_cov, tuple(blocks)
        self.sizes = [ob.points.n for ob in self.blocks]
        self.offs = np.concatenate([[0], np.cumsum(self.sizes)]).astype(int)
        self.n = int(self.offs[-1])
        self._lowered = _lowered
        self._noise = []
        for ob in self.blocks:
            if ob.b is None or not isinstance(ob.b, randvars.Normal):
                self._noise.append(None)
            elif ob.b.cov_diag is not None:
                self._noise.append(np.asarray(ob.b.cov_diag, dtype=np.double))
            else:
                self._noise.append(np.asarray(ob.b.cov, dtype=np.double).reshape(ob.points.n, ob.points.n))
        self._noise_diag = _noise_diag
        self.products = 0            # kernel entries evaluated so far (for the reports)

    @property
    def shape(self):
        return (self.n, self.n)

    def desc(self, i, j):
        return self._lowered(self.base, self.blocks[i].coeffs, self.blocks[j].coeffs)

    def matvec(self, V):
        V = np.asarray(V, dtype=np.double)
        V2 = V.reshape(self.n, -1)
        out = np.zeros_like(V2)
        for i, bi in enumerate(self.blocks):
            ri = slice(self.offs[i], self.offs[i + 1])
            for j, bj in enumerate(self.blocks):
                if self.sizes[i] == 0 or self.sizes[j] == 0:
                    continue
                out[ri] += _engine.kernel_matvec(self.ctx, self.desc(i, j), bi.points, bj.points,
                                                 np.ascontiguousarray(V2[self.offs[j]:self.offs[j + 1]]))
                self.products += self.sizes[i] * self.sizes[j]
            nz = self._noise[i]
            if nz is not None:
                out[ri] += nz[:, None] * V2[ri] if nz.ndim == 1 else nz @ V2[ri]
        return out.reshape(V.shape)

    __matmul__ = matvec

    def device_ok(self) -> bool:
        """Every noise term a diagonal: the product can run on resident operands."""
        return all(nz is None or nz.ndim == 1 for nz in self._noise)

    def matvec_dev(self, V, Q) -> None:
        """Q = G V on `DeviceVectors` (launches only)."""
        for i, bi in enumerate(self.blocks):
            if self.sizes[i] == 0:
                continue
            first = True
            for j, bj in enumerate(self.blocks):
                if self.sizes[j] == 0:
                    continue
                _engine.kernel_matvec_dev(self.ctx, self.desc(i, j), bi.points, bj.points, V, self.offs[j], Q, self.offs[i], not first)
                first = False
                self.products += self.sizes[i] * self.sizes[j] * V.m
            nz = self._noise[i]
            if nz is not None:
                Q.scale_rows_add(self.offs[i], V, nz)

    def diag(self):
        d = np.empty(self.n)
        for i, ob in enumerate(self.blocks):
            d[self.offs[i]:self.offs[i + 1]] = _engine.kernel_diag(self.ctx, self.desc(i, i))
            nz = self._noise[i]
            if nz is not None:
                d[self.offs[i]:self.offs[i + 1]] += nz if nz.ndim == 1 else np.diag(nz)
        return d

    def row(self, p: int):
        """Row p of G (n entries), evaluated on the device from one point against every block."""
        i = int(np.searchsorted(self.offs, p, side="right") - 1)
        l = p - self.offs[i]
        Xi = self.blocks[i].X.reshape(self.sizes[i], -1)
        one = _engine.Points(self.ctx, Xi[l:l + 1])
        r = np.empty(self.n)
        for j, bj in enumerate(self.blocks):
            if self.sizes[j]:
                r[self.offs[j]:self.offs[j + 1]] = _engine.kernel_matrix(self.ctx, self.desc(i, j), one, bj.points)[0]
        nz = self._noise[i]
        if nz is not None:
            if nz.ndim == 1:
                r[p] += nz[l]
            else:
                r[self.offs[i]:self.offs[i + 1]] += nz[l]
        return r


class PivotedCholeskyPreconditioner:
    """`M = L^T L + delta I`, `L` (r x n) from r greedy pivots of G; `M^{-1}` by Woodbury."""

    def __init__(self, G: GramProduct, rank: int, rtol: float = 1e-6):
        n = G.n
        d = G.diag().copy()
        d0 = float(np.max(d))
        L = np.zeros((min(rank, n), n))
        k = 0
        while k < L.shape[0]:
            p = int(np.argmax(d))
            if d[p] <= rtol * d0:
                break
            r = G.row(p)
            if k:
                r = r - L[:k, p] @ L[:k]          # the part of row p the earlier pivots already explain
            L[k] = r / np.sqrt(d[p])
            d = np.maximum(d - L[k] * L[k], 0.0)
            d[p] = 0.0
            k += 1
        self.L = L[:k]
        self.rank = k
        self.delta = max(float(np.mean(d)), 1e-12 * d0)
        if self.rank:
            S = self.delta * np.eye(self.rank) + self.L @ self.L.T
            self._chol = np.linalg.cholesky(S)
        else:
            self._chol = None

    def solve(self, R):
        if self._chol is None:
            return R / self.delta
        T = self.L @ R
        T = np.linalg.solve(self._chol.T, np.linalg.solve(self._chol, T))
        return (R - self.L.T @ T) / self.delta


def pcg(matvec, B, precond=None, X0=None, rtol: float = 1e-10, maxiter: int = 2000):
    """Preconditioned conjugate gradients for SPD `matvec`, all columns of B at once (one product per iteration for all of
    them, independent step lengths).  Returns (X, info) with info = {iterations, converged, rel_residual (per column)}."""
    B = np.asarray(B, dtype=np.double)
    vec = B.ndim == 1
    B2 = B.reshape(B.shape[0], -1)
    X = np.zeros_like(B2) if X0 is None else np.array(np.asarray(X0, dtype=np.double).reshape(B2.shape))
    R = B2 - matvec(X) if X0 is not None else B2.copy()
    bn = np.linalg.norm(B2, axis=0)
    bn[bn == 0.0] = 1.0
    Z = precond.solve(R) if precond is not None else R
    P = Z.copy()
    rz = np.sum(R * Z, axis=0)
    it, rel = 0, np.linalg.norm(R, axis=0) / bn
    while it < maxiter and np.any(rel > rtol):
        Q = matvec(P)
        pq = np.sum(P * Q, axis=0)
        active = (rel > rtol) & (pq > 0.0)
        alpha = np.where(active, rz / np.where(pq > 0.0, pq, 1.0), 0.0)
        X += alpha * P
        R -= alpha * Q
        Z = precond.solve(R) if precond is not None else R
        rz_new = np.sum(R * Z, axis=0)
        beta = np.where(active, rz_new / np.where(rz != 0.0, rz, 1.0), 0.0)
        P = Z + beta * P
        rz = rz_new
        rel = np.linalg.norm(R, axis=0) / bn
        it += 1
    info = {"iterations": it, "converged": bool(np.all(rel <= rtol)), "rel_residual": rel.copy()}
    return (X[:, 0] if vec else X.reshape(B.shape)), info


def pcg_device(G: "GramProduct", B, precond=None, X0=None, rtol: float = 1e-10, maxiter: int = 2000):
    """The same iteration as `pcg` with every operand resident on the device (round 6): `G.matvec_dev` for the product,
    `lpgp_pcg_step` for the rest.  Same return value."""
    B = np.asarray(B, dtype=np.double)
    vec = B.ndim == 1
    B2 = np.ascontiguousarray(B.reshape(B.shape[0], -1))
    n, m = B2.shape
    ctx = G.ctx
    DV = _engine.DeviceVectors
    X = DV(ctx, n, m, None if X0 is None else np.asarray(X0, dtype=np.double).reshape(n, m))
    R = DV(ctx, n, m, B2)
    Z, P, Q = DV(ctx, n, m), DV(ctx, n, m), DV(ctx, n, m)
    if X0 is not None:
        G.matvec_dev(X, Q)
        R.axpby(R, Q, -1.0)                       # R = B - G X0
    bn = np.linalg.norm(B2, axis=0)
    bn[bn == 0.0] = 1.0
    if precond is not None and precond.rank:
        Sinv = np.linalg.inv(precond._chol @ precond._chol.T)
        it_ = _engine.DevicePCG(ctx, n, m, precond.L, 0.5 * (Sinv + Sinv.T), precond.delta)
    else:
        it_ = _engine.DevicePCG(ctx, n, m, None, None, 1.0 if precond is None else precond.delta)
    rel = it_.start(R, Z, P, bn, rtol)
    it = 0
    while it < maxiter and np.any(rel > rtol):
        G.matvec_dev(P, Q)
        rel = it_.step(X, R, Z, P, Q, rtol)
        it += 1
    Xh = X.get()
    info = {"iterations": it, "converged": bool(np.all(rel <= rtol)), "rel_residual": rel.copy(), "device_resident": True}
    return (Xh[:, 0] if vec else Xh.reshape(B.shape)), info


class _MatrixFreeGram:
    """`gram` of a matrix-free posterior: the part of probnum's `LinearOperator` protocol that makes sense without a factor."""

    def __init__(self, gp: "MatrixFreeConditionalGaussianProcess"):
        self._gp = gp

    @property
    def shape(self):
        return self._gp._G.shape

    dtype = np.dtype(np.double)
    is_symmetric = True
    is_positive_definite = True

    @property
    def T(self):
        return self

    def __matmul__(self, V):
        return self._gp._G.matvec(V)

    def solve(self, B):
        X, info = self._gp._solve(B)
        return X

    def todense(self):
        raise NotImplementedError("a matrix-free Gram operator has no dense form (that is its point); use `@` and `solve`")

    cholesky = todense


class MatrixFreeConditionalGaussianProcess:
    """Posterior GP whose Gram matrix is never formed.  Built by `GaussianProcess.condition_on_observations` when
    `lp.config.matrix_free` is set or the number of observations exceeds `lp.config.matrix_free_above`."""

    def __init__(self, prior, blocks, warm_start=None):
        self._prior = prior
        self._blocks = tuple(blocks)
        self._ctx = _engine.default_context()
        self._G = GramProduct(self._ctx, prior.cov, self._blocks)
        self._precond = None
        self._weights = None
        self._warm = warm_start
        self.last_solve_info = None
        d = max(int(np.prod(prior.input_shape, dtype=int)), 1)
        self._test_coeffs = {(0,) * d: 1.0}

    # -- construction ---------------------------------------------------------------------
    @classmethod
    def from_observations(cls, prior, Y, X=None, *, L=None, b=None, previous=None):
        from ._gaussian_process import ConditionalGaussianProcess, _ObservationBlock
        Yf, Lf, bf, Xpts, coeffs, pred_mean = ConditionalGaussianProcess._preprocess_observations(prior=prior, Y=Y, X=X, L=L, b=b)
        ctx = _engine.default_context()
        block = _ObservationBlock(Yf, Lf, bf, Xpts, coeffs, Lf.device_points(ctx), pred_mean)
        old = () if previous is None else previous._blocks
        warm = None
        if previous is not None and previous._weights is not None:
            warm = np.concatenate([previous._weights, np.zeros(block.points.n)])
        return cls(prior, old + (block,), warm_start=warm)

    def condition_on_observations(self, Y, X=None, *, L=None, b=None):
        if any(v != 1.0 or any(mi) for mi, v in self._test_coeffs.items()):
            raise NotImplementedError("conditioning a transformed posterior is not supported")
        return MatrixFreeConditionalGaussianProcess.from_observations(self._prior, Y, X, L=L, b=b, previous=self)

    # -- reference surface ------------------------------------------------------------------
    @property
    def prior(self):
        return self._prior

    @property
    def input_shape(self):
        return self._prior.input_shape

    @property
    def input_ndim(self):
        return len(self._prior.input_shape)

    @property
    def output_shape(self):
        return ()

    @property
    def gram(self):
        return _MatrixFreeGram(self)

    def _preconditioner(self):
        if self._precond is None:
            rank = int(config.matrix_free_preconditioner_rank)
            self._precond = PivotedCholeskyPreconditioner(self._G, rank) if rank > 0 else None
        return self._precond

    def _solve(self, B, X0=None):
        if config.matrix_free_device_iteration and self._G.device_ok() and np.asarray(B).reshape(self._G.n, -1).shape[1] <= 256:
            X, info = pcg_device(self._G, B, self._preconditioner(), X0=X0, rtol=float(config.matrix_free_rtol),
                                 maxiter=int(config.matrix_free_maxiter))
        else:
            X, info = pcg(self._G.matvec, B, self._preconditioner(), X0=X0, rtol=float(config.matrix_free_rtol),
                          maxiter=int(config.matrix_free_maxiter))
        self.last_solve_info = info
        if not info["converged"]:
            raise np.linalg.LinAlgError(
                f"conjugate gradients did not reach rtol = {config.matrix_free_rtol:g} in {info['iterations']} iterations "
                f"(relative residual {float(np.max(info['rel_residual'])):.2e}): the Gram matrix is too ill-conditioned for the "
                "matrix-free path, or not positive definite")
        return X, info

    def _residual(self):
        return np.concatenate([ob.Y - ob.pred_mean for ob in self._blocks])

    @property
    def representer_weights(self):
        if self._weights is None:
            self._weights, _ = self._solve(self._residual(), X0=self._warm)
        return self._weights

    # -- prediction -------------------------------------------------------------------------
    def _flat(self, x):
        x = np.asarray(x, dtype=np.double)
        batch = x.shape[: x.ndim - self.input_ndim]
        d = max(int(np.prod(self.input_shape, dtype=int)), 1)
        return np.ascontiguousarray(x.reshape(-1, d)), batch

    def _prior_mean_at(self, X):
        m = self._prior.mean
        d = len(next(iter(self._test_coeffs)))
        if isinstance(m, functions.Constant):
            return np.full(X.shape[0], self._test_coeffs.get((0,) * d, 0.0) * float(m.value))
        if set(self._test_coeffs) == {(0,) * d} and self._test_coeffs[(0,) * d] == 1.0:
            return np.asarray(m(X if self.input_ndim else X[:, 0]), dtype=np.double).reshape(-1)
        # a read-out `L(posterior)` of a non-constant mean: closed-form derivatives or NotImplementedError, as on the dense path
        return np.asarray(functions.apply_coefficients(self._test_coeffs, m)(X if self.input_ndim else X[:, 0]), dtype=np.double).reshape(-1)

    def _with_test_operator(self, L) -> "MatrixFreeConditionalGaussianProcess":
        """`L(posterior)` (`_conditional.py:432-450`): the same observations, weights and preconditioner, read out through `L`
        (`LinearFunctionOperator.__call__` is isinstance-based and knew the dense posterior only)."""
        out = MatrixFreeConditionalGaussianProcess.__new__(MatrixFreeConditionalGaussianProcess)
        out.__dict__.update(self.__dict__)
        out._test_coeffs = covfuncs._compose(L.coefficients_dict(), self._test_coeffs)
        return out

    def __call__(self, x) -> randvars.Normal:
        X, batch = self._flat(x)
        if len(batch) != 1:
            raise ValueError("`__call__` needs inputs of shape (N,) + input_shape")
        return randvars.Normal(self.mean(x), self.cov.matrix(x))

    def _cross_desc(self, ob):
        return self._G._lowered(self._prior.cov, self._test_coeffs, ob.coeffs)

    def _mean_flat(self, X, pts):
        w = self.representer_weights
        out = self._prior_mean_at(X)
        for i, ob in enumerate(self._blocks):
            if ob.points.n:
                out = out + _engine.kernel_matvec(self._ctx, self._cross_desc(ob), pts, ob.points,
                                                  np.ascontiguousarray(w[self._G.offs[i]:self._G.offs[i + 1]]))
        return out

    def _cross_dense(self, pts):
        """K_Xx (n x m) on the host, block by block (m is a chunk of the prediction points)."""
        K = np.empty((self._G.n, pts.n))
        for i, ob in enumerate(self._blocks):
            if ob.points.n:
                K[self._G.offs[i]:self._G.offs[i + 1]] = _engine.kernel_matrix(self._ctx, self._cross_desc(ob), pts, ob.points).T
        return K

    def mean(self, x):
        X, batch = self._flat(x)
        return self._mean_flat(X, _engine.Points(self._ctx, X)).reshape(batch)

    def predict(self, x, *, return_var: bool = True):
        X, batch = self._flat(x)
        pts = _engine.Points(self._ctx, X)
        mean = self._mean_flat(X, pts).reshape(batch)
        if not return_var:
            return mean
        kxx = _engine.kernel_diag(self._ctx, self._G._lowered(self._prior.cov, self._test_coeffs, self._test_coeffs))
        var = np.empty(X.shape[0])
        chunk = int(config.matrix_free_rhs_chunk)
        for c0 in range(0, X.shape[0], chunk):
            sub = _engine.Points(self._ctx, X[c0:c0 + chunk])
            K = self._cross_dense(sub)
            S, _ = self._solve(K)
            var[c0:c0 + chunk] = kxx - np.sum(K * S, axis=0)
        return mean, var.reshape(batch)

    def var(self, x):
        return self.predict(x, return_var=True)[1]

    def std(self, x):
        return np.sqrt(np.maximum(self.var(x), 0.0))

    @property
    def cov(self):
        return _MatrixFreeCovariance(self)


class _MatrixFreeCovariance:
    def __init__(self, gp):
        self._gp = gp

    def matrix(self, x0, x1=None):
        gp = self._gp
        X0, _ = gp._flat(x0)
        X1 = X0 if x1 is None else gp._flat(x1)[0]
        P0 = _engine.Points(gp._ctx, X0)
        P1 = P0 if x1 is None else _engine.Points(gp._ctx, X1)
        kxx = _engine.kernel_matrix(gp._ctx, gp._G._lowered(gp._prior.cov, gp._test_coeffs, gp._test_coeffs), P0, P1)
        K0 = gp._cross_dense(P0)
        K1 = K0 if x1 is None else gp._cross_dense(P1)
        S, _ = gp._solve(K1)
        return _engine.gemm(gp._ctx, K0, S, transa=True, alpha=-1.0, beta=1.0, C=kxx)      # kxx - K0^T S on the device

    def __call__(self, x0, x1=None):
        if x1 is None:
            return self._gp.var(x0)
        return np.diag(self.matrix(x0, x1)) if np.shape(x0) == np.shape(x1) else self.matrix(x0, x1)
