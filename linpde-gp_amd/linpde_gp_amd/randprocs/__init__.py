from . import covfuncs
from ._gaussian_process import ConditionalGaussianProcess, GaussianProcess

__all__ = ["covfuncs", "GaussianProcess", "ConditionalGaussianProcess"]
