"""Run-time knobs of the MI355X path (the reference's only flag, `pn.config.block_triangular_solves`,
`linops/_block.py:8-14`, has no counterpart: triangular solves are always blocked here)."""

# Expected final number of observations of a chain of `condition_on_observations` calls.
# The resident Gram/factor buffer is allocated once with this capacity instead of growing
# (and being copied) with every appended block.  0 = grow on demand.
import os as _os0
gram_capacity_hint: int = int(_os0.environ.get("LPGP_GRAM_CAPACITY_HINT", 0))

# Gram blocks whose two point sets are tensor grids (`domains.TensorProductGrid`, e.g. from
# `Box.uniform_grid`) are assembled as sums of Kronecker products of 1-D kernel matrices
# (`lpgp_gram_assemble_grid`); False forces the generic per-entry evaluation.
use_grid_assembly: bool = True
# ... for grids of at least this many points.  Below, the per-entry kernel is the faster of the two: the Kronecker path costs eight
# launches of 1-D factor matrices and an expansion kernel that is latency-bound on a small block (a 32 x 32 grid: 72 us against 8;
# steps of 1 024 .. 9 216 grid points are 1.7 - 6.4 % faster entry by entry, at 16 384 points (c3) the two are equal, at 65 536 (c4)
# the expansion's 5.3 TB/s wins; MEASUREMENTS.md round 5); the values agree to rounding either way
# (`test_small_grids_take_the_per_entry_kernel_and_agree_with_the_kronecker_path`).  Env LPGP_GRID_MIN_POINTS overrides.
import os as _os
grid_assembly_min_points: int = int(_os.environ.get("LPGP_GRID_MIN_POINTS", 12000))

# When a Gram matrix that is not positive definite is reported.
# False (default): inside `condition_on_observations`, as in the reference -- its constructor evaluates the representer
# weights (`_conditional.py:44` first conditioning, `:280-282` `schur_update` on re-conditioning; `:83` passes
# `self.representer_weights` to `Mean`), i.e. the Cholesky factor is computed and `np.linalg.LinAlgError` raised before
# `condition_on_observations` returns.  Code that wraps the call in try/except (a jitter retry) works unchanged.  The
# status word is read back once per conditioning (one host synchronisation).
# True (opt-in, a throughput knob): `condition_on_observations` ENQUEUES assembly and factorisation and returns; the
# status is read by the first call that needs the factor (`predict`, `mean`, `cov`, `representer_weights`,
# `gram.cholesky()` ...), which raises `np.linalg.LinAlgError` on the object whose block failed and on every object
# conditioned on it; the objects before it stay usable (their part of the factor is untouched, the failed blocks are
# dropped).  The host runs ahead of the device over a chain of conditionings, and a `predict` that follows rides inside
# the last factorisation (`lpgp_potrf_predict_enqueue`).  The numbers are the same in both modes; only WHERE a failure
# surfaces differs from the reference.  `bench.py` opts in and says so in its line.
# Multi-GPU jobs always check inside `condition_on_observations` (the ranks agree on the status collectively).
lazy_factorization: bool = False

# A MEAN-only request (`u.mean(x)`) computes the marginal variance in the same pass and keeps it for the `u.std(x)` /
# `u.var(x)` on the same points that usually follows (notebook 0001 cell 22 calls `u.mean(grid)`, then `u.std(grid)`): the pair
# then costs one `predict` -- in lazy mode the fused factor-and-predict pipeline -- instead of a solve for the representer
# weights plus a second cross-covariance assembly and the forward substitution.  Off by default: a caller who never asks for
# the variance would pay the forward substitution of the cross-covariance for nothing.
# (Independently of this flag every posterior keeps its LAST prediction, so `mean(x)` after `predict(x)` is free.)
variance_with_mean: bool = False

# Matrix-free posteriors (`randprocs/_matrix_free.py`): the Gram matrix is never formed, every product re-evaluates its entries
# on the GPU (`lpgp_kernel_matvec`, the reference's KeOps slot, `diffops/_matern.py:112-135`) and solves are preconditioned
# conjugate gradients.  `matrix_free = True`: every first conditioning builds one; `matrix_free_above = N`: first conditionings
# on more than N observations do (190 000 points are a 289-GB dense Gram matrix: beyond one device).  0 / False: never.
matrix_free: bool = False
matrix_free_above: int = 0
matrix_free_preconditioner_rank: int = 200      # pivoted-Cholesky rank (0: plain CG)
matrix_free_rtol: float = 1e-10                  # relative residual of every CG solve
matrix_free_maxiter: int = 5000
matrix_free_rhs_chunk: int = 64                  # prediction points per block of variance solves
matrix_free_device_iteration: bool = True        # iterates / residuals / search directions resident in HBM, one iteration = launches only
                                                 # (`lpgp_pcg_step`, round 6); False: the host loop of round 5 (NumPy vector algebra around `lpgp_kernel_matvec`)

# Lazy mode: a block that was only assembled stays unfactored across FURTHER conditionings -- to be factored together with them,
# the prediction riding inside -- only if it has at least this many rows; smaller ones are factored when the next conditioning
# arrives (their kernels run while the host prepares it).
defer_min_rows: int = 4096
