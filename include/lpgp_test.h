/*
 * lpgp_test.h -- test hooks of the MI355X GP-posterior path.  NOT part of the product ABI (include/lpgp.h):
 * implemented in csrc/testhooks.hip, built into liblpgp_testhooks.so (which links against liblpgp.so) and loaded only
 * by tests/ (tests/_hooks.py) and scratch/.  Raw kernels on host buffers for unit tests and micro-benchmarks, the peak
 * probes, the host replay of the distributed tile enumeration.
 */
#ifndef LPGP_TEST_H
#define LPGP_TEST_H

#include "lpgp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HOST replay of the tile enumeration of a distributed trailing update (no GPU needed): the local tiles of rank
 * (my_r, my_c) of a pr x pc grid in rows [row_lo, T) x columns [col_lo, T) (GLOBAL tile indices, blocks of nbt tiles)
 * that lie on or below the diagonal, in launch order.  out (capacity cap pairs): (global tile row, global tile column)
 * per list entry; returns the number of tiles (< 0: error).                                          */
int  lpgp_test_stair_enumerate(int32_t pr, int32_t pc, int32_t my_r, int32_t my_c, int32_t nbt, int32_t T,
                               int32_t row_lo, int32_t col_lo, int32_t* out, int64_t cap);
/* C(m x n, col-major ldc) = beta*C + alpha * op(A) op(B); ta/tb: 0 => operand stored with
 * its non-contracted index fastest, 1 => contracted index (k) fastest.                  */
int  lpgp_test_gemm(lpgp_ctx* ctx, int32_t ta, int32_t tb, int32_t lower_only,
                    int64_t m, int64_t n, int64_t k, double alpha,
                    const double* A, int64_t lda, const double* B, int64_t ldb,
                    double beta, double* C, int64_t ldc, int32_t reps, double* ms_per_rep);
/* in-place 128x128 tile Cholesky + inverse: T (col-major 128x128) -> L, Linv            */
int  lpgp_test_potrf_tile(lpgp_ctx* ctx, double* T, double* Linv, int32_t* info);
/* the two in-place tile solves of the panel chain / forward substitution on host buffers, each with its one
 * step of iterative refinement (X0 = A Linv^T, X = X0 + (A - X0 L^T) Linv^T):
 *   which = 0:  X (rows x 128, col-major ldx = rows, rows a multiple of 128) <- X * L^{-T}
 *   which = 1:  V (128 x cols, col-major ld 128, cols a multiple of 128) <- L^{-1} * V
 * L, Linv: 128 x 128 column-major, lower triangular WITH ZEROS above the diagonal (as the tile Cholesky
 * stores them); Linv the fp64 inverse of L.                                                    */
int  lpgp_test_tile_step(lpgp_ctx* ctx, int32_t which, double* XV, int64_t n, const double* L,
                         const double* Linv, double* ms);
/* the fused panel step of the forward substitution on host buffers: V (nt * 128 rows x cols, col-major, ld = nt * 128,
 * cols a multiple of 128) <- Lblk^{-1} V with Lblk the nt x nt tile lower-triangular block (col-major, ld nt * 128, the
 * diagonal TILES with zeros above their diagonal) and Linv the nt explicit inverses of its diagonal tiles.
 * rows_form != 0: the same chain for rows, X (cols rows x nt * 128 columns, col-major, ld = cols) <- X Lblk^{-T}
 * (the panel solve of the multi-GPU factorisation and of a block append).                                       */
int  lpgp_test_panel_solve(lpgp_ctx* ctx, int32_t rows_form, double* V, int32_t nt, int64_t cols, const double* Lblk,
                           const double* Linv, double* ms);
/* diagnostics: histogram over the 8 XCDs of where the single workgroup of the tile Cholesky ran
 * since the last reset (the CU reservation of the update streams is built on it)           */
int  lpgp_debug_tile_xcc(lpgp_ctx* ctx, int32_t* out8, int32_t reset);
/* peak probes: fp64 MFMA issue loop and streaming write; returns TFLOP/s resp. GB/s     */
int  lpgp_probe_mfma_f64(lpgp_ctx* ctx, double* tflops);
int  lpgp_probe_hbm_write(lpgp_ctx* ctx, int64_t bytes, double* gbps);

/* leaves the sticky status word of `mat` as an enqueued factorisation that ended with `value` would (value < 0: a hand-over of
 * a resident kernel timed out): the error path of lpgp_mat_check without a broken device                          */
int  lpgp_test_force_status(lpgp_ctx* ctx, lpgp_mat* mat, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* LPGP_TEST_H */
