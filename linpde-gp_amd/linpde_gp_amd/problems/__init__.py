"""Problem definitions: the reference's `problems.pde` builders (API surface) and the
synthetic BASELINE workloads used by bench.py and the tests."""

from . import pde
from ._workloads import (
    Observation,
    Workload,
    analytic_solution,
    build_prior,
    condition_and_predict,
    heat_1d,
    heat_reference,
    operator_of,
    poisson_1d,
    poisson_2d,
    row_residual,
    scattered_2d,
    upload,
)

__all__ = [
    "pde", "Observation", "Workload", "build_prior", "condition_and_predict", "heat_1d", "heat_reference",
    "operator_of", "poisson_1d", "poisson_2d", "scattered_2d", "upload", "row_residual", "analytic_solution",
]
