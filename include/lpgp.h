/*
 * lpgp.h -- C ABI of the MI355X-native GP-posterior hot path of linpde-gp.
 *
 * Every entry point replaces a call site of the reference's two Python plug-in
 * protocols, cited as /root/reference/src/linpde_gp/<file>:<line>.
 * Plain pointers and sizes only; no torch types.  Host pointers are BORROWED for
 * the duration of a call; device state lives behind opaque handles that the caller
 * frees explicitly.  All calls are synchronous from the caller's point of view
 * unless stated otherwise.  Return value: 0 = ok, <0 = error (text in
 * lpgp_last_error()).  fp64 everywhere, C-order host arrays.
 */
#ifndef LPGP_H
#define LPGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPGP_MAXD 4      /* input dimension of a tensor-product kernel            */
#define LPGP_MAXT 256    /* terms of the expansion  sum_t c_t prod_d d^{n0} d'^{n1} k_d: every pair of second-order
                          * operators in four dimensions (15 x 15 multi-indices) fits                     */
#define LPGP_MAXG 16     /* summands of a sum kernel                              */

/* LPGP_MATERN_ISO: ISOTROPIC half-integer Matern over all d input dimensions,
 * k(x,x') = kappa_nu(|| sqrt(2 nu) (x - x') / lengthscale ||)  (probnum `Matern` with
 * input_shape (d,)); every dimension of the descriptor then carries this family and the same p.
 * It is not a product over dimensions: closed forms exist for at most ONE derivative per
 * argument (sum of the orders of n0 <= 1, of n1 <= 1), i.e. identity and directional derivatives:
 * `HalfIntegerMatern_Identity_DirectionalDerivative` (diffops/_matern.py:17-86) and
 * `HalfIntegerMatern_DirectionalDerivative_DirectionalDerivative` (:138-203).                */
enum lpgp_family { LPGP_MATERN_HALFINT = 1, LPGP_EXPQUAD = 2, LPGP_MATERN_ISO = 3 };

typedef struct lpgp_ctx lpgp_ctx;   /* one per process / per GPU                          */
typedef struct lpgp_pts lpgp_pts;   /* device-resident point set (n x d)                  */
typedef struct lpgp_mat lpgp_mat;   /* device-resident SPD matrix -> Cholesky factor      */
typedef struct lpgp_rhs lpgp_rhs;   /* device-resident n x m block of right-hand sides    */
typedef struct lpgp_dvec lpgp_dvec; /* device-resident n x m block of plain vectors (matrix-free path: no Gram matrix behind it) */
typedef struct lpgp_pcg lpgp_pcg;   /* state of preconditioned conjugate gradients on such blocks */

/* One term  coef * prod_d  d^{n0[d]}/dx_d^{n0[d]}  d^{n1[d]}/dx'_d^{n1[d]}  k_d(x_d, x'_d)
 * of `TensorProduct_LinDiffOp_LinDiffOp.__init__`
 * (randprocs/covfuncs/linfuncops/diffops/_tensor_product.py:38-67).                     */
typedef struct {
  double  coef;
  int32_t n0[LPGP_MAXD];
  int32_t n1[LPGP_MAXD];
} lpgp_term;

/* One scaled tensor-product kernel with both operators applied:
 *   scale * sum_t terms[t]      ( = L0 (scale * k_1 (x) ... (x) k_d) L1'^* ).
 * family/p/lengthscale per dimension: Matern nu = p + 1/2 (probnum Matern, scale factor
 * sqrt(2 nu)/lengthscale; diffops/_matern.py:17-639) or ExpQuad
 * exp(-(x-x')^2/(2 l^2)) (diffops/_expquad.py:12-433).                                   */
typedef struct {
  int32_t   d;
  int32_t   family[LPGP_MAXD];
  int32_t   p[LPGP_MAXD];
  double    lengthscale[LPGP_MAXD];
  double    scale;
  int32_t   nterms;
  lpgp_term terms[LPGP_MAXT];
} lpgp_kdesc;

/* ---- context ------------------------------------------------------------------------
 * THE CONTRACT IS ONE PROCESS PER GPU.  SURVEY.md section 8(b) sketched `lpgp_init(int ndev, const int* dev_ids, int pr, int pc,
 * lpgp_ctx**)` with "one process drives all 8 GPUs (single-process multi-device RCCL)".  This library deviates, deliberately:
 * lpgp_init takes ONE device, a context owns one GPU's streams, pools and communicator rank, and a multi-GPU job is N
 * processes that each call lpgp_init + lpgp_dist_init (rank, world, the RCCL unique id).  Why: (i) the task's launch contract is
 * one rank per GPU under `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE), and the driver's scaling bench starts the
 * ranks that way; (ii) a single host thread driving eight devices serialises eight panel chains of ~10 us launches through
 * one Python interpreter -- the chain is latency-bound, eight interpreters keep eight chains fed;
 * (iii) RCCL's group semantics and HIP's per-thread current device make single-process multi-device code a second code
 * path to test, for no data-path difference (the same xGMI copies).  The reference's calling convention -- ONE Python
 * process calling `condition_on_observations` (_conditional.py:253-294) -- is kept above the C ABI: `lp.spawn(n)`
 * (linpde_gp_amd/_spawn.py) starts one worker process per GPU and replays the caller's calls SPMD.    */
int  lpgp_init(int device, lpgp_ctx** ctx);
int  lpgp_finalize(lpgp_ctx* ctx);
const char* lpgp_last_error(void);
/* name (<= len bytes), compute units, HBM bytes of the device behind ctx */
int  lpgp_device_info(lpgp_ctx* ctx, char* name, int len, int* cus, int64_t* hbm_bytes);
int  lpgp_sync(lpgp_ctx* ctx);                       /* hipDeviceSynchronize */
/* tuning knobs (env LPGP_NB / LPGP_LOOKAHEAD give the defaults): panel width of the
 * blocked Cholesky (multiple of 128) and look-ahead on/off                              */
int  lpgp_set_option(lpgp_ctx* ctx, const char* key, int64_t value);
int  lpgp_get_option(lpgp_ctx* ctx, const char* key, int64_t* value);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI -----------------------------------
 * Pr x Pc process grid (rank = r * Pc + c), 2-D block-cyclic tiles: tile (i, j) of the padded matrix lives on
 * rank ((i / 4) % Pr, (j / 4) % Pc) (blocks of nb = 512 rows / columns) and ONLY there -- the factor is sharded,
 * never replicated (only its nb x nb diagonal blocks and the tile inverses are kept everywhere).  Assembly is local
 * (every rank evaluates the tiles it owns).  The rows below a factored panel travel from the Pr ranks of its process
 * column to every rank by point-to-point sends posted as one RCCL group (direct peer copies over the xGMI mesh);
 * every rank then updates its own tiles.  Solves stream the factor panel by panel against right-hand sides that
 * are sharded by column (csrc/dist.hip).
 * Rank 0 creates the id, the caller ships the 128 bytes to the other ranks (any control plane), every rank then
 * calls lpgp_dist_init before its first lpgp_mat_create.  lpgp_dist_set_grid (optional, the same on every rank: before
 * lpgp_dist_init / lpgp_dist_init_host, or later while no matrix / right-hand side is alive) fixes Pr x Pc; the default is Pr = world, Pc = 1: on the
 * full mesh of xGMI links every link then carries 1 / world of a panel.
 * All lpgp_* calls that touch a distributed matrix are COLLECTIVE: every rank makes them in the same order.   */
int  lpgp_dist_unique_id(char* out128);
int  lpgp_dist_set_grid(lpgp_ctx* ctx, int32_t pr, int32_t pc);
int  lpgp_dist_init(lpgp_ctx* ctx, int32_t rank, int32_t world, const char* uid128);
int  lpgp_dist_info(lpgp_ctx* ctx, int32_t* rank, int32_t* world);
int  lpgp_dist_grid(lpgp_ctx* ctx, int32_t* pr, int32_t* pc);
/* bytes this rank has sent / received in panel exchanges since lpgp_init (or the last call with reset != 0);
 * the time inside them is profiling slot LPGP_K_COMM                                                        */
int  lpgp_dist_stats(lpgp_ctx* ctx, double* bytes_sent, double* bytes_received, int32_t reset);
/* Link probe: what the panel exchanges can expect from the fabric, measured through the calls they use (RCCL
 * ncclSend / ncclRecv groups, or device-to-device copies into IPC-mapped windows) with HIP events on the panel stream.
 * out: world*world + world + 2 doubles, rates in GB/s (0 = not measured on this rank): [s*world + d] the ordered pair s -> d
 * alone; [world*world + s] one link of s while s sends to every peer at once; [world*world + world] total inbound rate
 * of this rank while every rank sends to every peer at once (the pattern of a Pr x 1 panel gather); [world*world + world + 1]
 * the bytes per message actually moved (the receive window of the direct-peer transport may cap `bytes`).  Collective.  The
 * reference has no counterpart (single process, host BLAS); it exists so that ONE run on an 8-GPU node can be held
 * against the communication model of DESIGN.md section 7 (bench.py: config.link_probe).                        */
int  lpgp_dist_link_probe(lpgp_ctx* ctx, int64_t bytes, int32_t reps, double* out);
/* Bring-up / test transport: the same distributed algorithms with every message staged through the host and
 * handed to a caller-supplied exchange (op 0: broadcast `bytes` bytes of buf from rank `root`; op 1: element-wise
 * max all-reduce of one int32 in buf; return 0 on success) instead of RCCL.  Ranks may then share ONE GPU (RCCL
 * refuses that), which is how tests/test_gpu_dist.py runs 2-, 3- and 4-rank jobs (2 x 2 grid included) on the
 * single-GPU box.  All arithmetic still runs on the device; it is not a data path of the product (bench.py never
 * selects it).                                                                                              */
typedef int (*lpgp_host_exchange_fn)(void* user, int32_t op, void* buf, int64_t bytes, int32_t root);
int  lpgp_dist_init_host(lpgp_ctx* ctx, int32_t rank, int32_t world, lpgp_host_exchange_fn fn, void* user);
/* Direct-peer transport, the alternative to RCCL: every rank exports a receive window in its HBM
 * (lpgp_dist_ipc_export: hipMalloc + hipIpcGetMemHandle, 64-byte handle out), the caller ships the handles (any control
 * plane), lpgp_dist_init_ipc maps every peer's window (hipIpcOpenMemHandle; handles = world x 64 bytes, rank-major).
 * A panel piece is then pushed by its root straight into the windows of all peers with device-to-device copies (one per
 * peer, every xGMI link of the root busy at once) -- no ring, no staging through the host; only the barriers between
 * the phases of an exchange (and the info all-reduce) use the caller's exchange (op 1 of lpgp_host_exchange_fn).
 * Several ranks may share one GPU, so tests/test_gpu_dist.py runs multi-rank jobs with a real device data path on
 * the single-GPU box.  Host-synchronous per exchange (the root drains its stream before the barrier): it does not
 * overlap with the look-ahead the way the RCCL group does.                                                    */
int  lpgp_dist_ipc_export(lpgp_ctx* ctx, int64_t window_bytes, char* handle64);
int  lpgp_dist_init_ipc(lpgp_ctx* ctx, int32_t rank, int32_t world, const char* handles,
                        lpgp_host_exchange_fn fn, void* user);

/* ---- point sets (X of `_EvaluationFunctional`, linfunctls/_evaluation.py:21-45) -----
 * X_host is consumed when lpgp_pts_create returns.  On a single GPU the upload of a small set (<= 64 KB) is asynchronous on
 * the panel stream (pinned staging ring), where every kernel that reads a point set runs, and lpgp_pts_destroy recycles the
 * device buffer: the reference hands NumPy arrays over per call, a small problem's step creates and drops half a dozen sets. */
int  lpgp_pts_create(lpgp_ctx* ctx, const double* X_host, int64_t n, int32_t d, lpgp_pts** out);
int  lpgp_pts_destroy(lpgp_pts* pts);

/* ---- Gram matrix: replaces `(L_i k L_j'^*).linop(X_i, X_j)` + todense()
 *      (crosscov/linfunctls/_evaluation.py:163-173, randvars/_covariance.py:197-224),
 *      the noise add (_conditional.py:392-394) and the BlockMatrix2x2 assembly
 *      (_conditional.py:275-281).
 *      The matrix is a sequence of observation blocks (one per conditioning step), in
 *      conditioning order.  Storage: lower triangle, column-major; every block is padded
 *      to a multiple of 128 rows with an identity tail (internal; all sizes and vectors in
 *      this API are LOGICAL, i.e. without padding).                                      */
int  lpgp_mat_create(lpgp_ctx* ctx, int64_t capacity_hint, lpgp_mat** out);
int  lpgp_mat_destroy(lpgp_mat* mat);
/* declare the next observation block of n rows; returns its index (>= 0) or < 0       */
int  lpgp_mat_add_block(lpgp_ctx* ctx, lpgp_mat* mat, int64_t n);
/* Undo the last lpgp_mat_add_block whose block has NOT been factored (a failed lpgp_potrf leaves it
 * so): the matrix is again what it was before the block was declared -- the leading factor, its tile
 * inverses and the earlier blocks are untouched by a failed append (its phases only write the new
 * rows / columns).  This is how a failed `condition_on_observations` leaves the object it was called on
 * intact, as in the reference (a failed `BlockMatrix2x2` construction has no side effect on the old
 * operator, linops/_block.py:84-130).                                                          */
int  lpgp_mat_pop_block(lpgp_ctx* ctx, lpgp_mat* mat);
/* VIEW on the leading nblocks observation blocks (nblocks >= 1, all of them factored): every solve /
 * prediction call then sees exactly the factor an earlier conditioning produced -- a block append never
 * touches the leading part of the factor -- so an earlier posterior object stays usable after a later
 * one has extended the shared matrix (reference semantics: posteriors are immutable values,
 * _conditional.py:253-294).  nblocks = -1: all blocks.  While a strict prefix is in view the matrix
 * cannot be extended or assembled into (branching = lpgp_mat_clone).                            */
int  lpgp_mat_set_view(lpgp_ctx* ctx, lpgp_mat* mat, int32_t nblocks);
int32_t lpgp_mat_num_blocks(const lpgp_mat* mat);     /* blocks in view */
int32_t lpgp_mat_num_blocks_total(const lpgp_mat* mat);
/* independent copy of the leading nblocks (factored) blocks: a second conditioning of an object that
 * has already been extended continues on its own copy (device-to-device, 8 n^2 bytes)          */
int  lpgp_mat_clone(lpgp_ctx* ctx, const lpgp_mat* mat, int32_t nblocks, lpgp_mat** out);
int64_t lpgp_mat_size(const lpgp_mat* mat);           /* logical size = sum of block sizes (of the view) */
int64_t lpgp_mat_padded_size(const lpgp_mat* mat);    /* internal padded size              */
/* block (bi, bj), bi >= bj  <-  sum_g (kd[g])(X0, X1) with X0 the points of block bi
 * (operator on argument 0) and X1 those of block bj (operator on argument 1).  For
 * bi == bj pass X1 == NULL: only the lower triangle is evaluated and stored.           */
int  lpgp_gram_assemble(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups,
                        const lpgp_pts* X0, const lpgp_pts* X1,
                        lpgp_mat* mat, int32_t bi, int32_t bj);
/* The same block when its point sets are TENSOR GRIDS: rows = grid F0[0] x ... x F0[d-1] and
 * columns = grid F1[0] x ... x F1[d-1] of 1-D point sets (C order, last factor fastest; F1 ==
 * NULL for bi == bj).  The block is then a sum of Kronecker products of 1-D kernel matrices:
 * O(T d n^2) kernel evaluations + an HBM-write-bound expansion instead of N^2 evaluations.
 * Replaces the Kronecker `linop` of `TensorProduct` / `TensorProduct_LinDiffOp_LinDiffOp`
 * (covfuncs/_tensor_product.py:64-82, diffops/_tensor_product.py:140-156).                  */
int  lpgp_gram_assemble_grid(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups,
                             const lpgp_pts* const* F0, const lpgp_pts* const* F1,
                             lpgp_mat* mat, int32_t bi, int32_t bj);
/* 1 if lpgp_gram_assemble_grid holds this sum (its tables of terms and of distinct 1-D matrices are fixed-size kernel
 * arguments: 48 terms over all summands, 16 distinct 1-D matrices per dimension; product-form kernels only), 0 if the
 * caller must assemble the block entry-wise from the flattened grids (lpgp_gram_assemble).                              */
int  lpgp_kron_fits(const lpgp_kdesc* kd, int32_t ngroups);
/* diagonal of block bi += v_host[i] (v_host may be NULL) + scalar                       */
int  lpgp_mat_add_diag(lpgp_ctx* ctx, lpgp_mat* mat, int32_t bi, const double* v_host, double scalar);
/* diagonal block bi += B_host (n_bi x n_bi, C-order, symmetric)                          */
int  lpgp_mat_add_dense(lpgp_ctx* ctx, lpgp_mat* mat, int32_t bi, const double* B_host);
/* dense copy-out (n x n, C-order, n = lpgp_mat_size).  what = 0: symmetric Gram as
 * assembled (only valid while nothing is factored), 1: lower Cholesky factor.           */
int  lpgp_mat_to_host(lpgp_ctx* ctx, lpgp_mat* mat, int32_t what, double* out_host);
/* diagonal of the Cholesky factor (n = lpgp_mat_size doubles): `LinearOperator.det / logabsdet` of the Gram operator
 * (probnum protocol behind `ConditionalGaussianProcess.gram`, _conditional.py:92-94) without collecting the factor:
 * log det G = 2 sum_i log L_ii.  Local in a multi-GPU job (the diagonal blocks are replicated).                      */
int  lpgp_mat_factor_diag(lpgp_ctx* ctx, lpgp_mat* mat, double* out_host);

/* ---- factor + solve: replaces `gram.solve`, `LinearOperator.cholesky`
 *      (_conditional.py:44,108) and the Schur-complement append
 *      `BlockMatrix2x2.schur_update/_cholesky` (linops/_block.py:192-242).
 *      Factors every block that is not factored yet; blocks factored by an earlier call
 *      are kept (block append).  info: 0 ok, k > 0 => leading minor of (padded) order k
 *      is not positive definite.                                                         */
int  lpgp_potrf(lpgp_ctx* ctx, lpgp_mat* mat, int32_t* info);
/* The same factorisation ENQUEUED: no host synchronisation, the status stays on the device (one sticky word per
 * matrix: the first non-positive pivot).  NOT the reference's order of events -- its constructor evaluates the
 * representer weights (_conditional.py:44, :83, :280-282), so a Gram matrix that is not positive definite raises inside
 * `condition_on_observations`; lpgp_potrf reproduces that and is what the host package uses by default.  The enqueued
 * form is the opt-in throughput mode (`lp.config.lazy_factorization = True`): the host runs ahead of the device over a
 * chain of conditionings (c3: four boundary blocks of ~0.2 ms each, half of it host time behind a status read-back),
 * and the failure surfaces at the first use of the factor.  Single GPU only (the multi-GPU factorisation agrees on its
 * status collectively).
 * lpgp_mat_check waits for the panel stream and returns the status of everything enqueued since the last check:
 * info = 0 ok; k > 0: the leading minor of (padded) order k is not positive definite, *block = index of the
 * observation block it lies in -- the factor of blocks 0 .. *block - 1 is intact.  lpgp_mat_truncate drops the blocks
 * from `nblocks` on (the full view must be current) and clears the status: the matrix is again what it was before they
 * were declared, exactly as after lpgp_mat_pop_block.                                                                 */
int  lpgp_potrf_enqueue(lpgp_ctx* ctx, lpgp_mat* mat);
/* ONE CONDITIONING IN ONE CALL (round 4) -- `ConditionalGaussianProcess.from_observations / condition_on_observations`
 * (_conditional.py:253-294, :392-394): declare the new block of n rows, assemble its block row of the Gram matrix --
 * row[j] describes (L_new k L_j'^*)(X_new, X_j) for every earlier block j = 0 .. nrow - 2, row[nrow - 1] the diagonal block
 * (L_new k L_new'^*)(X_new, X_new) (X1 = NULL there; F0 / F1 non-NULL: both point sets are tensor grids, the Kronecker path
 * of lpgp_gram_assemble_grid) --, add the noise b.cov (a scalar variance, or noise_diag[n], or noise_dense[n x n], at most
 * one of them), and factor: lazy == 0 factors and returns the status in *info (the reference's order of events); lazy == 1
 * enqueues the factorisation (lpgp_potrf_enqueue; status by lpgp_mat_check); lazy == 2 assembles only and leaves the
 * factorisation to whichever call needs the factor first -- lpgp_potrf / lpgp_potrf_enqueue, or lpgp_potrf_predict, inside
 * which the prediction rides; further lazy == 2 conditionings may follow first: all blocks that were only assembled are then
 * factored TOGETHER, one factorisation from the first unfactored column on (a chain of small conditionings costs one panel
 * chain instead of one per block; a large block in the MIDDLE of a chain -- c5 -- is factored with the prediction riding inside it).  On an error -- and on info != 0 -- the block is dropped again (lpgp_mat_pop_block):
 * the matrix is what it was before the call.  The same launches as the separate calls, in the same order; what it saves
 * is host time: nrow + 3 calls through the binding per conditioning (at N_tot ~ 1 000 a conditioning is bound by the
 * host), and the synchronisation the noise upload of lpgp_mat_add_diag needs for its borrowed host vector (here the
 * vector is staged into the matrix's own storage on an idle stream).  Single GPU and multi-GPU alike (multi-GPU: lazy
 * must be 0).                                                                                                      */
typedef struct lpgp_cond_block {
  const lpgp_kdesc* kd;
  int32_t ngroups;
  const lpgp_pts* X1;               /* points of block j; NULL for the diagonal block (and for the Kronecker path) */
  const lpgp_pts* const* F0;        /* Kronecker path: kd[0].d factor point sets of the new block ... */
  const lpgp_pts* const* F1;        /* ... and of block j (NULL for the diagonal block) */
} lpgp_cond_block;
int  lpgp_mat_condition(lpgp_ctx* ctx, lpgp_mat* mat, int64_t n, const lpgp_pts* X_new, const lpgp_cond_block* row, int32_t nrow,
                        double noise_scalar, const double* noise_diag, const double* noise_dense, int32_t lazy, int32_t* info);
int  lpgp_mat_check(lpgp_ctx* ctx, lpgp_mat* mat, int32_t* info, int32_t* block);
int  lpgp_mat_truncate(lpgp_ctx* ctx, lpgp_mat* mat, int32_t nblocks);
/* x = G^{-1} b for nrhs right-hand sides, b_host (n x nrhs, column-major, in/out)      */
int  lpgp_potrs(lpgp_ctx* ctx, lpgp_mat* mat, double* b_host, int64_t nrhs);
/* representer weights w = G^{-1} r; keeps w resident for lpgp_predict; w_host may be
 * NULL (_conditional.py:96-110).  Single GPU: ONE resident launch per direction whose workgroups hand the solution over
 * block by block (csrc/trsv.hip; option "trsv_resident" / LPGP_TRSV_RESIDENT, 0: one launch per 128-row tile as in rounds
 * 1-5); r_host and w_host pass through pinned staging, the call returns with w_host written.  A hand-over that does not
 * arrive within its bound (a dispatch order the kernel does not expect) is reported as an error, not a hang.            */
int  lpgp_solve_weights(lpgp_ctx* ctx, lpgp_mat* mat, const double* r_host, double* w_host);
/* hands the residual r = Y - L[m] - b.mean (_conditional.py:44) to the factored matrix
 * WITHOUT solving for the weights: lpgp_predict with both mean and variance requested then
 * forms the mean as V^T z with z = L^{-1} r (forward substitution only, overlapped with the
 * solve of the cross-covariance).  Invalidated by lpgp_mat_add_block / lpgp_potrf.        */
int  lpgp_mat_set_residual(lpgp_ctx* ctx, lpgp_mat* mat, const double* r_host);

/* ---- prediction: replaces `PriorPredictiveCrossCovariance._evaluate`
 *      (_conditional.py:140-153), `Mean._evaluate` (:193-197) and
 *      `CovarianceFunction._evaluate` (:223-231).                                       */
/* STREAMS (for callers of the C API that keep work asynchronous): everything is ordered on the context's panel stream.
 * lpgp_mat_set_residual (single GPU) copies r_host into pinned staging and enqueues the upload on the panel stream WITHOUT
 * waiting for it -- r_host is consumed when it returns, every reader of the residual (lpgp_predict, lpgp_potrf_predict) is
 * ordered behind the upload there.  lpgp_mat_condition's noise-vector upload uses an idle side stream ONLY while an enqueued
 * factorisation is in flight on the panel stream (`lpgp_potrf_enqueue` without `lpgp_mat_check` yet) -- then nothing else
 * can be reading the weights buffer it is staged in, because every entry point that reads it (lpgp_solve_weights, lpgp_potrs,
 * lpgp_predict) returns only after its device work has completed.                                                     */
int  lpgp_rhs_create(lpgp_ctx* ctx, const lpgp_mat* mat, int64_t m, lpgp_rhs** out);
int  lpgp_rhs_destroy(lpgp_rhs* rhs);
/* rows of block bi of K_Xx <- sum_g (kd[g])(X_obs, X_test)   (n_bi x m)                 */
int  lpgp_cross_assemble(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups,
                         const lpgp_pts* X_obs, const lpgp_pts* X_test,
                         lpgp_rhs* rhs, const lpgp_mat* mat, int32_t bi);
/* The same for ALL observation blocks in one call (round 5): blocks[bi] = {descriptor, observation points} of block bi (kd == NULL:
 * the block has no cross-covariance with the prediction points, its rows stay zero).  Consecutive blocks that share a descriptor
 * -- value observations on several boundary pieces -- are assembled by ONE launch (a table of point sets in the kernel
 * arguments), as lpgp_mat_condition does for a block row of the Gram matrix (_conditional.py:140-153, :270).        */
typedef struct lpgp_cross_block {
  const lpgp_kdesc* kd;
  int32_t ngroups;
  const lpgp_pts* X_obs;
} lpgp_cross_block;
int  lpgp_cross_assemble_row(lpgp_ctx* ctx, const lpgp_cross_block* blocks, int32_t nblocks, const lpgp_pts* X_test,
                             lpgp_rhs* rhs, const lpgp_mat* mat);
/* mean_host[j] = prior_mean_host[j] + K_Xx[:, j] . w     (prior_mean_host may be NULL)
 * var_host[j]  = kxx_host[j] - || L^{-1} K_Xx[:, j] ||^2  (skipped if var_host == NULL)
 * K_Xx is overwritten by V = L^{-1} K_Xx when the variance is requested.  If no weights
 * are resident but a residual is (lpgp_mat_set_residual) and both outputs are requested,
 * the mean is formed as prior_mean + V[:, j] . (L^{-1} r) instead.                       */
int  lpgp_predict(lpgp_ctx* ctx, lpgp_mat* mat, lpgp_rhs* K,
                  const double* prior_mean_host, const double* kxx_host,
                  double* mean_host, double* var_host);
/* FACTOR WHAT IS NOT FACTORED YET AND PREDICT, IN ONE PIPELINE (round 5).  Same outputs as lpgp_predict with mean and
 * variance requested and a residual resident (lpgp_mat_set_residual): V = L^{-1} K_Xx, mean = m + V^T (L^{-1} r), var =
 * k_xx - colsum(V^2) -- but the forward substitution of K_Xx (and of the residual, the spare column) rides INSIDE the
 * factorisation of the blocks declared since the last factorisation: every panel of the factor hands its columns to the
 * substitution the moment its chain is complete, and the substitution's panel steps run on a stream of their own,
 * never on the factorisation's critical path (potrf.hip, `ride_panel`).  The algebra is `BlockMatrix2x2.L_A_inv_B`
 * (linops/_block.py:203-207) applied to the cross-covariance `PriorPredictiveCrossCovariance._evaluate`
 * (_conditional.py:140-153) panel by panel; the values are those of lpgp_potrf_enqueue + lpgp_predict, kernel for kernel.
 * The factorisation's status is not RETURNED here: lpgp_mat_check afterwards (a Gram matrix that is not positive definite
 * yields a prediction computed on garbage, to be discarded) -- the status word travels back with mean and variance (one
 * read-back into pinned memory, one wait), so that lpgp_mat_check costs no device round trip of its own.  If everything is
 * factored already this is lpgp_predict.  Single GPU.
 * For a panel the RESIDENT chain factors (chain.hip) and a right-hand side of at most 96 x 32 columns, the substitution's
 * panel step follows the factor workgroup through the chain's device flags instead of waiting for the chain kernel
 * (`panel_chain_v_kernel`; option `ride_vchain_max_wgs`, env LPGP_RIDE_VCHAIN, 0: off).                                */
int  lpgp_potrf_predict(lpgp_ctx* ctx, lpgp_mat* mat, lpgp_rhs* K, const double* prior_mean_host,
                        const double* kxx_host, double* mean_host, double* var_host);
/* V <- L^{-1} V  (forward substitution on all m columns)                                */
int  lpgp_trsm_lower(lpgp_ctx* ctx, lpgp_mat* mat, lpgp_rhs* V);
/* out_host (ma x mb, C-order) = A^T B   (covariance update V0^T V1)                     */
int  lpgp_rhs_inner(lpgp_ctx* ctx, lpgp_rhs* A, lpgp_rhs* B, double* out_host);
/* logical rows, n x m C-order                                                           */
int  lpgp_rhs_to_host(lpgp_ctx* ctx, const lpgp_mat* mat, lpgp_rhs* rhs, double* out_host);
/* *out_new = a NEW block  A[:, :ma] B  with B_host (ma x m, C-order; ma = columns of A) sharing A's row layout (lpgp_rhs_destroy):
 * the product behind the posterior covariance AS A LINEAR OPERATOR -- `(k_xx - kLas_x0 @ gram.solve(kLas_x1.T)) @ V`,
 * `_conditional.py:245-251` -- as  k(x0, x1) V - V0^T (V1 V)  with V0 = L^{-1} K_Xx0, V1 = L^{-1} K_Xx1 resident: the n0 x n1
 * matrix never exists.  Single GPU.                                                                                   */
int  lpgp_rhs_matmul(lpgp_ctx* ctx, const lpgp_rhs* A, const double* B_host, int64_t m, lpgp_rhs** out_new);
/* C (m x n) = alpha op(A) op(B) + beta C on HOST arrays (all C-order; op(A) is m x k: A is m x k, or k x m with transa != 0;
 * likewise B) through the fp64 MFMA kernel: the dense products the reference leaves to NumPy around the path --
 * `A @ Sigma0`, `crosscov @ A.T`, `Sigma0 - crosscov.T @ gain` of the finite-dimensional conditioning (`randvars/_normal.py:8-71`),
 * `L @ L.T` of `gram.todense()`.  beta == 0: C is not read.  Single GPU.                                           */
int  lpgp_gemm_host(lpgp_ctx* ctx, int32_t transa, int32_t transb, int64_t m, int64_t n, int64_t k, double alpha,
                    const double* A_host, const double* B_host, double beta, double* C_host);
/* diag of sum_g (kd[g])(x, x): a constant for the stationary kernels supported here     */
int  lpgp_kernel_diag(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups, double* out_value);

/* dense block (n0 x n1, C-order) of sum_g (kd[g])(X0, X1): replaces
 * `CovarianceFunction.__call__/matrix` (probnum protocol; call sites
 * crosscov/linfunctls/_evaluation.py:79,160,170)                                        */
int  lpgp_kernel_matrix(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups,
                        const lpgp_pts* X0, const lpgp_pts* X1, double* out_host);

/* matrix-free product  out (n0 x nrhs, C-order) = [sum_g (kd[g])(X0, X1)] v   with v (n1 x nrhs,
 * C-order): every kernel entry is evaluated on the fly, the n0 x n1 matrix is never formed.
 * Replaces the `_keops_lazy_tensor` hooks (diffops/_matern.py:112-135,231-264,
 * experiments/cpu.py:214-229) behind `CovarianceFunction.linop(x0, x1) @ v`.               */
int  lpgp_kernel_matvec(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups,
                        const lpgp_pts* X0, const lpgp_pts* X1,
                        const double* v_host, int64_t nrhs, double* out_host);

/* ---- the same product with every operand RESIDENT (round 6) --------------------------------------------------------
 * The reference's KeOps lazy tensors keep their operands on the device and probnum's `LinearOperator.solve` iterates on the
 * product (diffops/_matern.py:112-135, experiments/cpu.py:214-229); until round 5 every iteration here went through host
 * vectors.  An `lpgp_dvec` is an n x m block of vectors in HBM (one column per right-hand side); host arrays are C-order.   */
int  lpgp_dvec_create(lpgp_ctx* ctx, int64_t n, int64_t m, lpgp_dvec** out);              /* zero-filled */
int  lpgp_dvec_destroy(lpgp_dvec* d);
int  lpgp_dvec_set(lpgp_ctx* ctx, lpgp_dvec* d, const double* host);
int  lpgp_dvec_get(lpgp_ctx* ctx, const lpgp_dvec* d, double* host);
/* out = a + s b (same shapes; out may be a or b)                                                                      */
int  lpgp_dvec_axpby(lpgp_ctx* ctx, lpgp_dvec* out, const lpgp_dvec* a, const lpgp_dvec* b, double s);
/* Y[off : off + n, :] += diag(d_host) V[off : off + n, :]: the noise diagonal of a Gram block (`_conditional.py:392-394`) */
int  lpgp_dvec_scale_rows_add(lpgp_ctx* ctx, lpgp_dvec* Y, int64_t y_off, const lpgp_dvec* V, int64_t v_off, const double* d_host, int64_t n);
/* Y[y_off : y_off + n0, :] (accumulate != 0: +=) [sum_g (kd[g])(X0, X1)] V[v_off : v_off + n1, :] -- launches only        */
int  lpgp_kernel_matvec_dev(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups, const lpgp_pts* X0, const lpgp_pts* X1,
                            const lpgp_dvec* V, int64_t v_off, lpgp_dvec* Y, int64_t y_off, int32_t accumulate);
/* Preconditioned conjugate gradients for all m columns at once, one iteration = launches only: column dots in a fixed order of
 * summation, step lengths formed on the device, preconditioner M^{-1} = (I - L^T S^{-1} L) / delta (L: rank x n pivoted-Cholesky
 * rows, Sinv = (delta I + L L^T)^{-1}, both handed over once; rank 0: M = delta I).  The caller forms Q = G P between the steps
 * (lpgp_kernel_matvec_dev per block pair + noise terms); each call returns the m relative residuals (8 m bytes read back).  */
int  lpgp_pcg_create(lpgp_ctx* ctx, int64_t n, int64_t m, int32_t rank, const double* L_host, const double* Sinv_host, double delta, lpgp_pcg** out);
int  lpgp_pcg_destroy(lpgp_pcg* p);
/* R = B - G X0 on entry; Z = M^{-1} R, P = Z                                                                           */
int  lpgp_pcg_start(lpgp_ctx* ctx, lpgp_pcg* p, const lpgp_dvec* R, lpgp_dvec* Z, lpgp_dvec* P, const double* bnorm_host, double rtol, double* rel_host);
/* Q = G P on entry; X += alpha P, R -= alpha Q, Z = M^{-1} R, P = Z + beta P (columns already below rtol are left alone) */
int  lpgp_pcg_step(lpgp_ctx* ctx, lpgp_pcg* p, lpgp_dvec* X, lpgp_dvec* R, lpgp_dvec* Z, lpgp_dvec* P, const lpgp_dvec* Q, double rtol, double* rel_host);

/* ---- measurement: HIP-event timing of the hot kernels on their own streams ---------- */
enum lpgp_kernel_id { LPGP_K_ASSEMBLE = 0, LPGP_K_SYRK = 1 /* rank-nb trailing update */, LPGP_K_GEMM = 2,
                      LPGP_K_POTRF_TILE = 3, LPGP_K_TRSM = 4,
                      LPGP_K_SYRK_PANEL = 5 /* rank-128 triangular update inside a panel */,
                      LPGP_K_GEMM_SMALL = 6 /* any product small enough for the 64x64-tile kernel */,
                      LPGP_K_MATVEC = 7 /* matrix-free kernel product */,
                      LPGP_K_SYRK_AHEAD = 8 /* look-ahead half of the trailing update (next panel's columns) */,
                      LPGP_K_ASSEMBLE_GRID = 9 /* tensor-grid (Kronecker) expansion kernels; LPGP_K_ASSEMBLE = per-entry kernel */,
                      LPGP_K_PANEL = 10 /* fused panel factorisation / fused substitution tile steps */,
                      LPGP_K_COMM = 11 /* multi-GPU panel exchanges (bytes = sent + received by this rank) */,
                      LPGP_K_COUNT = 12 };
/* mask: bit k enables HIP-event bracketing of kernel id k (0 = off, -1 = all)          */
int  lpgp_profile_enable(lpgp_ctx* ctx, int32_t mask);
int  lpgp_profile_reset(lpgp_ctx* ctx);
/* accumulated over all launches since reset: device milliseconds (HIP events on the
 * launching stream), launch count, algorithmic flops, algorithmic bytes                */
int  lpgp_profile_get(lpgp_ctx* ctx, int32_t kernel_id, double* ms, int64_t* launches,
                      double* flops, double* bytes);

/* The raw-kernel test entry points, peak probes and diagnostics of earlier rounds (lpgp_test_*, lpgp_probe_*,
 * lpgp_debug_*) are NOT part of this library any more: include/lpgp_test.h, liblpgp_testhooks.so (tests only).      */

#ifdef __cplusplus
}
#endif
#endif /* LPGP_H */
