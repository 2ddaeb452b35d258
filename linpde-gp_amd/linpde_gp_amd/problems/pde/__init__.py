"""PDE problem builders -- host mirror of `linpde_gp.problems.pde`
(`_linear_pde.py`, `_bvp.py:15-171`, `_poisson.py:14-134`, `_heat.py:16-144`).
Thin containers producing the differential operator, right-hand side, boundary parts and
(where known) the analytic solution; no arithmetic of the hot path."""

from __future__ import annotations

import dataclasses
from collections.abc import Sequence

import numpy as np

from ... import domains, functions, linfuncops
from ...linfuncops import diffops


@dataclasses.dataclass(frozen=True)
class LinearPDE:
    domain: domains.Domain
    diffop: linfuncops.LinearFunctionOperator
    rhs: functions.Function

    def __post_init__(self):
        if self.diffop.input_domain_shape != self.domain.shape:
            raise ValueError("operator and domain shapes do not match")


class PoissonEquation(LinearPDE):
    def __init__(self, domain, rhs=None, alpha: float = 1.0):
        domain = domains.asdomain(domain)
        if rhs is None:
            rhs = functions.Zero(domain.shape)
        object.__setattr__(self, "_alpha", float(alpha))
        super().__init__(domain=domain, diffop=-float(alpha) * diffops.Laplacian(domain.shape), rhs=rhs)

    @property
    def alpha(self):
        return self._alpha


class HeatEquation(LinearPDE):
    def __init__(self, domain, rhs=None, alpha: float = 1.0):
        domain = domains.asdomain(domain)
        if rhs is None:
            rhs = functions.Zero(domain.shape)
        object.__setattr__(self, "_alpha", float(alpha))
        super().__init__(domain=domain, diffop=diffops.HeatOperator(domain.shape, alpha=float(alpha)), rhs=rhs)

    @property
    def alpha(self):
        return self._alpha


class BoundaryCondition:
    def __init__(self, boundary, operator, values):
        self._boundary = domains.asdomain(boundary)
        if operator.input_domain_shape != self._boundary.shape:
            raise ValueError(
                "The shape of the domain of the boundary operator's input function is not equal to the "
                f"shape of the given domain object ({operator.input_domain_shape} != {self._boundary.shape}).")
        self._operator = operator
        if not isinstance(values, functions.Function):
            values = functions.Constant(self._boundary.shape, values)
        if values.input_shape != self._boundary.shape:
            raise ValueError("boundary values have the wrong input shape")
        self._values = values

    @property
    def boundary(self):
        return self._boundary

    @property
    def operator(self):
        return self._operator

    @property
    def values(self):
        return self._values


class DirichletBoundaryCondition(BoundaryCondition):
    def __init__(self, boundary, values):
        boundary = domains.asdomain(boundary)
        super().__init__(boundary, linfuncops.Identity(boundary.shape), values)


def get_1d_dirichlet_boundary_observations(dirichlet_bcs):
    """`_bvp.py:75-87`: the two end points of an interval and the prescribed values."""
    if len(dirichlet_bcs) != 2 or not all(isinstance(bc.boundary, domains.Point) for bc in dirichlet_bcs):
        raise ValueError("expected exactly two point boundary conditions")
    X_bc = np.asarray([float(bc.boundary) for bc in dirichlet_bcs])
    Y_bc = np.asarray([float(bc.values(np.asarray(x))) for bc, x in zip(dirichlet_bcs, X_bc)])
    return X_bc, Y_bc


@dataclasses.dataclass(frozen=True)
class BoundaryValueProblem:
    pde: LinearPDE
    boundary_conditions: Sequence
    solution: functions.Function | None = None

    @property
    def domain(self):
        return self.pde.domain


@dataclasses.dataclass(frozen=True)
class InitialBoundaryValueProblem(BoundaryValueProblem):
    initial_condition: DirichletBoundaryCondition | None = None

    @property
    def initial_domain(self):
        """{t0} x spatial domain (`_bvp.py:139-171`)."""
        t0 = self.pde.domain[0][0]
        return domains.CartesianProduct(domains.Point(t0), self.pde.domain[1])


class Solution_PoissonEquation_DirichletProblem_1D_RHSConstant(functions.Function):
    """-alpha u'' = rhs on [l, r], u(l) = ul, u(r) = ur  (`_poisson.py:98-134`)."""

    def __init__(self, domain, rhs, boundary_values, alpha=1.0):
        super().__init__(input_shape=(), output_shape=())
        self._l, self._r = tuple(domain)
        self._rhs, self._alpha = float(rhs), float(alpha)
        self._ul, self._ur = (float(v) for v in boundary_values)

    def _evaluate(self, x):
        l, r = self._l, self._r
        lin = self._ul + (self._ur - self._ul) * (x - l) / (r - l)
        return lin + self._rhs / (2.0 * self._alpha) * (x - l) * (r - x)


class PoissonEquationDirichletProblem(BoundaryValueProblem):
    def __init__(self, domain, *, rhs=None, alpha: float = 1.0, boundary_values=None, solution=None):
        pde = PoissonEquation(domain, rhs=rhs, alpha=alpha)
        if boundary_values is None:
            boundary_values = functions.Zero(pde.domain.shape)
        if pde.domain.shape == ():
            if not isinstance(pde.domain, domains.Interval):
                raise TypeError("In the scalar case, we only support Interval domains.")
            if isinstance(boundary_values, functions.Function):
                a, b = pde.domain
                boundary_values = (float(boundary_values(np.asarray(a))), float(boundary_values(np.asarray(b))))
            boundary_values = np.asarray(boundary_values, dtype=np.double)
            if solution is None and isinstance(pde.rhs, functions.Constant):
                solution = Solution_PoissonEquation_DirichletProblem_1D_RHSConstant(
                    pde.domain, rhs=float(pde.rhs.value), boundary_values=boundary_values, alpha=alpha)
        if isinstance(boundary_values, functions.Function):
            bcs = tuple(_restricted_bc(part, boundary_values) for part in pde.domain.boundary)
        else:
            vals = np.asarray(boundary_values, dtype=np.double)
            bcs = tuple(DirichletBoundaryCondition(part, v) for part, v in zip(pde.domain.boundary, vals))
        super().__init__(pde=pde, boundary_conditions=bcs, solution=solution)


def _restricted_bc(part, values: functions.Function):
    return DirichletBoundaryCondition(part, values)


class TruncatedSineSeries(functions.Function):
    """sum_k c_k sin((k+1) pi (x - l) / (r - l))  (`functions/_fourier.py:11`)."""

    def __init__(self, domain, coefficients):
        super().__init__(input_shape=(), output_shape=())
        self._domain = domains.asdomain(domain)
        self._coefficients = np.asarray(coefficients, dtype=np.double)

    @property
    def domain(self):
        return self._domain

    @property
    def coefficients(self):
        return self._coefficients

    def _evaluate(self, x):
        l, r = tuple(self._domain)
        k = np.arange(1, self._coefficients.size + 1)
        return np.sum(self._coefficients * np.sin(k * np.pi * (x[..., None] - l) / (r - l)), axis=-1)


class Solution_HeatEquation_DirichletProblem_1D_InitialTruncatedSineSeries_BoundaryZero(functions.Function):
    """`_heat.py:96-144`: each sine mode decays with exp(-alpha ((k+1) pi / (r-l))^2 (t - t0))."""

    def __init__(self, t0, spatial_domain, initial_values: TruncatedSineSeries, alpha):
        super().__init__(input_shape=(2,), output_shape=())
        self._t0, self._alpha = float(t0), float(alpha)
        self._l, self._r = tuple(spatial_domain)
        self._c = initial_values.coefficients

    def _evaluate(self, tx):
        t, x = tx[..., 0], tx[..., 1]
        k = np.arange(1, self._c.size + 1)
        lam = k * np.pi / (self._r - self._l)
        return np.sum(self._c * np.exp(-self._alpha * lam**2 * (t[..., None] - self._t0))
                      * np.sin(lam * (x[..., None] - self._l)), axis=-1)


class HeatEquationDirichletProblem(InitialBoundaryValueProblem):
    def __init__(self, t0, spatial_domain, T=float("inf"), rhs=None, alpha=1.0, initial_values=None, solution=None):
        spatial_domain = domains.asdomain(spatial_domain)
        domain = domains.CartesianProduct(domains.Interval(t0, T), spatial_domain)
        pde = HeatEquation(domain, rhs=rhs, alpha=alpha)
        if initial_values is None:
            initial_values = functions.Zero(spatial_domain.shape)
        if initial_values.input_shape != spatial_domain.shape or initial_values.output_shape != ():
            raise ValueError("initial values must be a scalar function on the spatial domain")
        initial_condition = DirichletBoundaryCondition(spatial_domain, initial_values)
        bcs = tuple(
            DirichletBoundaryCondition(domains.CartesianProduct(domain[0], part), np.zeros(()))
            for part in spatial_domain.boundary)
        if solution is None:
            if isinstance(initial_values, functions.Zero):
                solution = functions.Zero(domain.shape)
            elif (isinstance(spatial_domain, domains.Interval) and isinstance(initial_values, TruncatedSineSeries)
                    and initial_values.domain == spatial_domain):
                solution = Solution_HeatEquation_DirichletProblem_1D_InitialTruncatedSineSeries_BoundaryZero(
                    t0, spatial_domain, initial_values, alpha)
        super().__init__(pde=pde, boundary_conditions=bcs, solution=solution, initial_condition=initial_condition)


__all__ = [
    "LinearPDE", "PoissonEquation", "HeatEquation", "BoundaryCondition", "DirichletBoundaryCondition",
    "BoundaryValueProblem", "InitialBoundaryValueProblem", "PoissonEquationDirichletProblem",
    "HeatEquationDirichletProblem", "TruncatedSineSeries", "get_1d_dirichlet_boundary_observations",
]
