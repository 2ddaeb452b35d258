"""Run-time knobs of the MI355X path (the reference's only flag, `pn.config.block_triangular_solves`,
`linops/_block.py:8-14`, has no counterpart: triangular solves are always blocked here)."""

# Expected final number of observations of a chain of `condition_on_observations` calls.
# The resident Gram/factor buffer is allocated once with this capacity instead of growing
# (and being copied) with every appended block.  0 = grow on demand.
gram_capacity_hint: int = 0

# Gram blocks whose two point sets are tensor grids (`domains.TensorProductGrid`, e.g. from
# `Box.uniform_grid`) are assembled as sums of Kronecker products of 1-D kernel matrices
# (`lpgp_gram_assemble_grid`); False forces the generic per-entry evaluation.
use_grid_assembly: bool = True

# When a Gram matrix that is not positive definite is reported.
# True (default): as in the reference, whose Cholesky factor is a `functools.cached_property` evaluated at first use
# (`_conditional.py:92`, `linops/_block.py:203`) -- `condition_on_observations` enqueues assembly and factorisation and
# returns; `np.linalg.LinAlgError` is raised by the first call that needs the factor (`predict`, `mean`, `cov`,
# `representer_weights`, `gram.cholesky()` ...), on the object whose block failed and on every object conditioned on it;
# the objects before it stay usable (their part of the factor is untouched, the failed blocks are dropped).  The host
# runs ahead of the device over a chain of conditionings instead of waiting for a status word after each.
# False: the status is read back inside `condition_on_observations`, which then raises itself (rounds 1-3).
# Multi-GPU jobs always check inside `condition_on_observations` (the ranks agree on the status collectively).
lazy_factorization: bool = True
