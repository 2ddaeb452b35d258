"""CPU oracle for the GP-posterior hot path of linpde-gp  --  TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy restatement of the reference algorithm
(`/root/reference/src/linpde_gp`, cited per function as file:line).  It is the
*checker* for the HIP path, never the product:

* only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
  import it;
* nothing under `linpde-gp_amd/` imports it, and the product raises if the HIP
  extension is missing (there is no CPU fallback).

PARITY UNPINNED by reference fixtures: the reference holds no golden vectors for
this path, cannot be imported here (`probnum`, `jax`, `pykeops`
are absent, no network) and has no native code to compile.  The oracle is
therefore pinned by the independent checks the reference's own tests use:

* kernel derivatives against symbolic differentiation (SymPy) evaluated in
  50-digit mpmath on the reference's test grid shapes
  (`tests/.../diffops/test_diffops.py:15-42` uses JAX autodiff the same way);
* the polynomial tables against the values derived from
  `covfuncs/linfuncops/diffops/_matern.py:613-639` that SURVEY.md §8(a) A1 lists;
* iterative (Schur/block) conditioning == one-shot dense conditioning
  (`tests/linpde_gp/randprocs/test_posterior_gp.py:152-178`);
* block Cholesky == dense Cholesky (`tests/linpde_gp/linops/test_symmetric_block.py`);
* analytic PDE solutions (`problems/pde/_poisson.py:98-134`).
"""

from . import covfuncs, gp, polynomials  # noqa: F401
