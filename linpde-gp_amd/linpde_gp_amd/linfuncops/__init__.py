from . import diffops
from ._linfuncop import Identity, LinearFunctionOperator
from .diffops import LinearDifferentialOperator

__all__ = ["diffops", "Identity", "LinearFunctionOperator", "LinearDifferentialOperator"]
