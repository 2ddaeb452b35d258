"""`randvars`: `Normal` / `Constant` (`_normal.py`) and the `Covariance` classes (`_covariance.py`)."""

from ._covariance import ArrayCovariance, Covariance, LinearOperatorCovariance, aslinop
from ._normal import Constant, Normal, asrandvar, condition_normal_on_observations

__all__ = ["Normal", "Constant", "asrandvar", "condition_normal_on_observations",
           "Covariance", "ArrayCovariance", "LinearOperatorCovariance", "aslinop"]
