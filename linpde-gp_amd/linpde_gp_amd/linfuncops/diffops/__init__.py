from ._coefficients import MultiIndex, PartialDerivativeCoefficients
from ._operators import (
    Derivative,
    DirectionalDerivative,
    HeatOperator,
    Laplacian,
    LinearDifferentialOperator,
    PartialDerivative,
    ScaledLinearDifferentialOperator,
    SpatialLaplacian,
    TimeDerivative,
    WeightedLaplacian,
)

__all__ = [
    "MultiIndex", "PartialDerivativeCoefficients", "LinearDifferentialOperator",
    "PartialDerivative", "TimeDerivative", "Derivative", "DirectionalDerivative",
    "WeightedLaplacian", "Laplacian", "SpatialLaplacian", "HeatOperator",
    "ScaledLinearDifferentialOperator",
]
