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
