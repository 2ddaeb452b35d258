"""linpde_gp_amd -- MI355X-native GP-posterior hot path of linpde-gp.

Python host mirror of the reference's operator interface
(`GaussianProcess.condition_on_observations` / `LinearFunctional` /
`LinearDifferentialOperator`) over hand-written HIP kernels in liblpgp.so
(C ABI: include/lpgp.h).  No PyTorch, no CPU fallback: importing the package needs the
built extension, and any evaluation needs a visible MI355X.
"""

__version__ = "0.1.0"

from . import _lib  # noqa: F401  (fails loudly when liblpgp.so is missing)
from . import config  # noqa: E402
from ._engine import to_device  # noqa: E402
from ._spawn import spawn  # noqa: E402
from . import domains, functions, linfuncops, linfunctls, problems, randprocs, randvars  # noqa: E402
from .randprocs import ConditionalGaussianProcess, GaussianProcess  # noqa: E402

__all__ = [
    "domains", "functions", "linfuncops", "linfunctls", "problems", "randprocs", "randvars",
    "GaussianProcess", "ConditionalGaussianProcess", "to_device", "config", "spawn",
]
