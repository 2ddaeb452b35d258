// Internal declarations shared by the HIP translation units of liblpgp.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "lpgp.h"
#include "lpgp_desc.h"

namespace lpgp {

constexpr int TILE = 128;          // base tile: potrf_tile block, GEMM block tile, padding unit

// ---- 2-D block-cyclic distribution ------------------------------------------------------
// Tiles of 128 are dealt in blocks of nbt tiles (nb = 512 columns): global tile g belongs to block g / nbt, owned by
// member (g / nbt) % P of its process row / column.  A member stores its tiles in increasing global order.
struct Cyc {
  int32_t P = 1, me = 0, nbt = 4;
};
__host__ __device__ inline int cyc_owner(const Cyc& c, int g) { return (g / c.nbt) % c.P; }
// number of tiles with global index < g that `me` owns == local index of global tile g if owned
__host__ __device__ inline int cyc_before(const Cyc& c, int g) {
  const int B = g / c.nbt, rem = g - B * c.nbt;
  return ((B + c.P - 1 - c.me) / c.P) * c.nbt + ((B % c.P) == c.me ? rem : 0);
}
__host__ __device__ inline int cyc_l2g(const Cyc& c, int l) {
  const int lb = l / c.nbt;
  return (lb * c.P + c.me) * c.nbt + (l - lb * c.nbt);
}
// global padded index -> local index, or -1 if another member owns it
__host__ __device__ inline int64_t cyc_local(const Cyc& c, int64_t gi) {
  if (c.P == 1) return gi;                       // single member: local == global (no divisions on the single-GPU path)
  const int g = (int)(gi / TILE);
  if (cyc_owner(c, g) != c.me) return -1;
  return (int64_t)cyc_before(c, g) * TILE + (gi - (int64_t)g * TILE);
}
// where a rank keeps element (row, column) of the global padded matrix: (row cyclic over Pr, column over Pc)
struct Layout2D {
  Cyc rows, cols;
};



#define LPGP_HIP(expr)                                                              \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      ::lpgp::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,                \
                        hipGetErrorString(_e));                                     \
      return -1;                                                                    \
    }                                                                               \
  } while (0)

#define LPGP_TRY_RC(expr)         \
  do {                            \
    int _rc = (expr);             \
    if (_rc != 0) return _rc;     \
  } while (0)
#define LPGP_TRY(expr) LPGP_TRY_RC(expr)

// ---- profiling -------------------------------------------------------------------------
struct ProfSlot {
  double ms = 0.0;
  int64_t launches = 0;
  double flops = 0.0;
  double bytes = 0.0;
};

struct PendingEvent {
  hipEvent_t e0, e1;
  int kernel;
};

}  // namespace lpgp

struct lpgp_ctx {
  int device = 0;
  int cus = 0;
  hipStream_t s_main = nullptr;    // panel / critical-path stream (high priority)
  hipStream_t s_upd = nullptr;     // trailing-update stream (all CUs but `reserve`, default 8)
  hipStream_t s_upd_narrow = nullptr;  // same with `reserve_narrow` CUs (default 64) left to the panel chain
  hipStream_t s_upd_all = nullptr;     // unmasked update stream: the blocked solves have no whole-CU kernel to protect
  hipStream_t s_outer = nullptr;       // rank-nb_outer updates (a1), (b) of the factorisation (masked like s_upd)
  int single_stream = 0;               // LPGP_SINGLE_STREAM: all of the above alias s_main (ranks sharing one GPU in tests)
  hipEvent_t ev_outer[2] = {nullptr, nullptr};
  hipEvent_t ev_outer_fact[2] = {nullptr, nullptr};
  hipEvent_t ev_outer_a1[2] = {nullptr, nullptr};
  int64_t nb_outer = 2048;             // far columns are updated once per nb_outer columns (0 or <= nb: every panel) ...
  int nb_outer_min_tiles = 192;        // ... while more than this many tile columns remain
  int reserve_narrow = 64;
  hipEvent_t ev_chain_pre = nullptr;   // recorded on the panel stream right in front of a resident chain launch: a kernel that follows the chain through
                                       // its flags (panel_chain_v_kernel) is not dispatched before the chain kernel itself can be
  hipEvent_t ev_ride[4] = {nullptr, nullptr, nullptr, nullptr};   // ride-along substitution (potrf_predict_blocked): panel hand-overs [0, 1], fork / join [2, 3]
  int ride_stream = 1 + 8 * 7;         // ... runs on (first + 8 * second stream; potrf.hip): 0 s_outer, 1 s_upd_all, 2 s_upd_narrow, 3 the panel stream, 4 s_upd, 7 none
  int ride_old_ungated = 1;            // ... the steps of old panels (block append) are not held back by the gate
  int ride_occ3 = 1;                   // ... its updates may use the three-workgroups-per-CU kernel
  int append_split = 0;                // (OFF: measured flat, c3 50.3-50.8 either way, profiles/r06_append_split_ab.txt) block append: the last old panel's update of the new block split into the first new panel's columns (panel stream) and the
  int append_split_min_tiles = 16;     // remainder (update stream, under the first new chain), for new blocks of at least this many tile rows (LPGP_APPEND_SPLIT)
  hipEvent_t ev_append[2] = {nullptr, nullptr};
  int ride_b_on_ride = 0;              // ... the factorisation's remainder updates queue on the substitution's stream once its gate is open (LPGP_RIDE_B_ON_RIDE)
  int ride_aug = 0;                    // ... or, where the matrix has room for it, as ROWS of the matrix being factored (potrf.hip: augmented form; LPGP_RIDE_AUG)
  // resident panel chain (chain.hip): panels of four tiles with at most this many tile rows below them run their whole chain in
  // ONE launch whose workgroups hand over through device flags (-1: never)
  int chain_resident_max_rows = 32;
  int chain_ahead = 1;                 // resident chain: the look-ahead update by the previous panel rides in front of the NEXT panel's chain (one launch per panel
                                       // where two chains follow each other; LPGP_CHAIN_AHEAD=0: a launch of its own, round 5)
  int chain_ahead_min_rows = 12;       // ... where at least this many tile rows lie below that panel (LPGP_CHAIN_AHEAD_MIN_ROWS)
  int chain_resident2_max_rows = 0;    // ... and panels with MORE rows below (up to this many tile rows) as TWO launches that talk through the same flags: factor + in-block
                                       // workgroups on the panel stream, the rows below -- 16 rows and 68 KB of LDS per workgroup, two per CU -- on an idle masked stream.
                                       // 0: never -- the default: measured SLOWER (round 6: c2 7.8 -> 8.9 ms, c3 50.2 -> 53): at 68 KB a row workgroup shares its CU with an
                                       // update workgroup and runs at half speed; the 152 KB of the one-launch form are what keeps a CU to itself (MEASUREMENTS.md)
  hipEvent_t ev_chain_rows = nullptr;
  int trsv_resident = 1;               // single right-hand side: one resident launch per direction (trsv.hip); 0: one launch per tile (rounds 1-5)
  int* d_chain_flags = nullptr;        // ring of flag slots (zeroed; a launch zeroes the slot half a ring ahead)
  int64_t chain_launches = 0;
  int* chain_last_slot = nullptr;      // flag slot and panel of the last chain launch: the substitution's panel step that follows it
  int chain_last_p0 = -1;              // through the same flags (panel_chain_v_kernel) ...
  int ride_vchain_pre = 1;             // ... dispatched only once the chain kernel itself can be (ev_chain_pre)
  int ride_vchain_max_wgs = 96;        // ... for right-hand sides of at most this many 32-column workgroups (they wait ON the chip, one per CU; 0: never)
  int ride_same_stream_max_tiles = 0;  // ... on the panel stream itself for factors of at most this many tile rows
  int64_t ride_outer_rows = 2048;      // ... two-level form: rows below an outer block of this many rows are updated once per block (0: every panel updates all rows below) ...
  int ride_outer_min_tiles = 64;       // ... from this many tile rows on
  int ride_max_tiles = 384;            // factors of at least this many tile rows: factorisation and substitution back to back instead (potrf.hip)
  int ride_gate_pct = -1;              // (-1: by size, potrf.hip)              // ... its steps are held back until at most this percentage of the tile rows is left to factor (>= 100: released at once)
  hipEvent_t ev_panel[2] = {nullptr, nullptr};
  hipEvent_t ev_upd[2] = {nullptr, nullptr};
  int64_t nb = 512;                // panel width of the blocked Cholesky
  int64_t nb_outer_solve = 4096;   // forward substitution: rows below an outer block of this many rows are updated once per block (two-level scheme, potrf.hip); 0: plain right-looking
  int scoped_gather = 1;           // Pr, Pc > 1 grids: a panel's rows go only to the process row / column whose updates read them (0: to everyone, rounds 1-3)
  int panel_lds_extra = 0;         // bytes of LDS a fused panel launch asks for beyond its own 69 632 (set by the two-level forward substitution around its launches)
  int panel_exclusive = 1;         // two-level forward substitution: its panel chains wait for the tail of the long update instead of slipping into it (0: rounds' 4 first behaviour)
  int fused_ahead_min_us = 800;    // ... while the remainder update is estimated at least this long (the fused launch shares its CUs with the update for most of the update's duration)
  int fused_ahead = 1;             // forward substitution: the look-ahead update rides in front of the next fused panel chain (one launch; 0: a launch of its own, rounds 1-3)
  int small_ring2 = 32;            // rank-128 in-panel updates of at least this many 128-tiles run on the two-stage ring of the 64 x 64 kernel (four workgroups per CU); 0: never
  int nb_outer_solve_min_tiles = 384;  // ... from this many tile rows on (c4; measured no gain at c3 / c5 sizes)
  int64_t nb_solve = 0;            // panel width of the blocked forward substitution (0: by size, see trsm_lower_blocked)
  int64_t nb_big = 0;              // optional wider panels while more than nb_big_min_tiles tile rows remain (0 = off; measured: no gain at c3)
  int nb_big_min_tiles = 96;
  int lookahead = 1;
  // estimated duration of one tile step of the panel chain (factorisation / forward substitution) and of
  // the per-panel rest, in microseconds: decides whether the remainder update is released with the panel
  // (update-bound) or after the look-ahead half (chain-bound)
  double chain_us_tile = 150.0, solve_chain_us_tile = 30.0, chain_us_fixed = 80.0;   // (factorisation: re-swept after the tile solves got their refinement step, scratch/sweep_chain.sh; forward substitution: its fused panel chain takes 117 us per 4 tile rows + 65 us of look-ahead update at c3 -- with the factorisation's 150 us per tile row the second half of the c3 prediction held every remainder update back behind its look-ahead half, 100 us of idle update stream per panel: 23.3 -> 22.7 ms, profiles/r03_solve_chain_estimate.txt)
  int min_supertiles = 128;        // GEMM grid: shrink the super-tile edge until there are this many
  int dense_tiles = 1;             // GEMM grid: dense XCD-balanced tile enumeration (0: legacy super-tile dealing)
  int gemm_band = 8;               // GEMM grid: tile rows per band of the dense enumeration (an XCD works on band x 64/band tiles at a time)
  int fused_solve = 1;             // forward substitution: one launch per panel of <= 512 rows (panel_solve_kernel); 0: a tile solve and an update per tile
  int asm_fast = 1;                // per-entry assembly: descriptors of the common shapes (D <= 2, one group, <= 2 parity classes, degrees <= 4) on the specialised kernel (assemble_fast_kernel; bit-identical to the generic one)
  int kron_wide = 1;               // Kronecker expansion with 16-byte stores where the fast extent is even (kron2w_kernel)
  int asm_batch = 1;               // blocks of a block row that share a descriptor are assembled in one launch (assemble.hip: launch_assemble_batch)
  int asm_ct = 4;                  // assemble_fast_kernel: column tiles per workgroup, at most (LPGP_ASM_CT)
  int asm_factors = 0;             // per-entry assembly / matrix-free product: exponentials of Matern dimensions from per-point factors (eval_entries.h);
                                   // +13 % on the kernel, ~4x the rounding noise of the entries (two exps and a product instead of one exp): off by default
  int gemm3_fact = 0;              // ... inside the FACTORISATION only if set: beside the panel chain the third resident workgroup costs the chain what it gains the update (c3: condition phase 33.6 -> 34.1 ms with it, predict phase 22.9 -> 22.6 ms: the forward substitution keeps it)
  double gemm3_margin = 2.0;       // ... and, inside the factorisation / forward substitution, only while the remainder update is estimated to take this many times longer than the panel chain beside it
  int gemm3 = 768;                 // GEMM / SYRK launches (A not transposed) with at least this many 128 x 128 tiles use the three-workgroups-per-CU kernel (gemm3_f64_kernel); 0: never
  int small_tiles_max = 256;       // GEMM launches with at most this many 128x128 tiles use the 64x64-tile kernel
  // workspace
  // descriptor ring: an assembly launch copies its lowered descriptor into a pinned host slot,
  // from there (truly asynchronously) into the slot's device copy; the slot is reused only
  // after the event recorded behind its kernel has fired -- no stream sync per assembly call
  static constexpr int DESC_RING = 8;
  struct DescSlot { lpgp::DevDesc* h = nullptr; lpgp::DevDesc* d = nullptr; hipEvent_t done = nullptr; bool used = false; };
  DescSlot desc_ring[DESC_RING];
  int desc_next = 0;
  int* d_info = nullptr;           // potrf info word
  int* d_info_cur = nullptr;       // the word the running factorisation reports to (d_info, or the matrix's own sticky word)
  int* h_info_pinned = nullptr;    // pinned mirror (multi-GPU: a device-to-host copy into pageable memory would BLOCK behind a collective that waits for a dead peer)
  double* d_tmp = nullptr;         // small scratch (vectors)
  int64_t tmp_cap = 0;
  double* h_stage = nullptr;       // pinned host staging for the results of a prediction: ONE asynchronous device-to-host copy and
  double* h_stage_r = nullptr;     // ... and for the residual on its way up (lpgp_mat_set_residual: no wait at all)
  int64_t stage_r_cap = 0;
  hipEvent_t ev_stage_r = nullptr;
  int64_t stage_cap = 0;           // one wait per call (a copy into pageable memory is a blocking round trip of its own, ~25 us each)
  // caching allocator: freed device buffers are kept for reuse (a hipMalloc/hipFree pair
  // of a multi-GB Gram matrix costs more than the factorisation of a small problem)
  struct PoolBuf { void* p; size_t bytes; };
  std::vector<PoolBuf> pool;
  // Point sets handed over per call (the reference's calling convention: NumPy arrays) come and go with every step of a small
  // problem: their device buffers are recycled here (hipMalloc + hipFree cost ~50 us a pair, and hipFree waits for the device),
  // their uploads go through a ring of pinned slots on the panel stream -- where every kernel that reads a point set runs --
  // without a wait (lpgp_pts_create / lpgp_pts_destroy; single GPU)
  std::vector<PoolBuf> pts_pool;
  static constexpr int PTS_SLOTS = 8;
  static constexpr size_t PTS_SLOT_BYTES = 64 << 10;
  struct PtsSlot { void* h = nullptr; hipEvent_t done = nullptr; bool used = false; };
  PtsSlot pts_ring[PTS_SLOTS];
  int pts_next = 0;
  // multi-GPU (one process per GPU): Pr x Pc process grid, rank = r * Pc + c; 2-D block-cyclic tiles
  int rank = 0, world = 1;
  int pr = 1, pc = 1;
  int grid_set = 0;                // lpgp_dist_set_grid called (else a default grid is chosen at init)
  int dist_broken = 0;             // the communicator was aborted after a failure inside a collective call
  int live_mats = 0;               // matrices / right-hand sides alive: their storage was laid out for the CURRENT grid (lpgp_dist_set_grid)
  int dist_bcast = 0;              // panel exchanges as one ncclBroadcast per piece instead of the point-to-point group (LPGP_DIST_COLLECTIVE=bcast)
  double comm_bytes_sent = 0.0, comm_bytes_recv = 0.0;
  hipStream_t s_comm = nullptr;    // panel exchange (RCCL point-to-point group calls)
  hipEvent_t ev_comm[2] = {nullptr, nullptr};
  void* nccl_comm = nullptr;       // ncclComm_t
  void* nccl_comm_bulk = nullptr;  // second communicator (ncclCommSplit of the first): the bulk of a split panel gather travels on it,
                                   // beside the small exchanges of the panel chain (operations on ONE communicator are serialised)
  hipEvent_t ev_tail[2] = {nullptr, nullptr};   // the bulk part of the gather into panel buffer 0 / 1 has landed
  hipEvent_t ev_rows[2] = {nullptr, nullptr};   // the rows below panel i are solved (the bulk gather may read them)
  double dist_chain_us_comm = 120.0;   // per-panel communication on the chain of a multi-GPU factorisation (diagonal-block broadcast + head gather), for the chain-bound / update-bound decision (LPGP_DIST_CHAIN_US_COMM)
  int split_gather = 1;            // P x 1 grids with look-ahead: gather the next diagonal block's rows first, the rest off the chain
  double* d_pack_bulk = nullptr;   // pack / receive buffer of the bulk part
  size_t pack_bulk_cap = 0;
  lpgp_host_exchange_fn host_xfer = nullptr;   // test transport (lpgp_dist_init_host): panels staged through the host
  void* host_xfer_user = nullptr;
  // direct-peer transport (lpgp_dist_init_ipc): every rank owns a receive window in HBM, mapped into every peer by
  // IPC handle; a root pushes its pieces straight into the peers' windows (device-to-device copies over xGMI / inside
  // the GPU), the host exchange above only carries the barriers between the phases of an exchange
  double* ipc_window = nullptr;               // this rank's window
  size_t ipc_window_doubles = 0;              // used as two halves: [0]: exchanges of the panel stream, [1]: bulk exchanges of the exchange stream
  hipEvent_t ev_ipc[2] = {nullptr, nullptr};  // completion of the last copy-out of each half (the half may be pushed into again after it)
  int ipc_copyout_pending[2] = {0, 0};
  std::vector<double*> ipc_peer;              // window of rank r as mapped here (own window for r == rank)
  bool ipc() const { return !ipc_peer.empty(); }
  bool distributed() const { return nccl_comm != nullptr || host_xfer != nullptr || dist_broken; }
  double* d_pack = nullptr;        // packed panel pieces: [own piece | pieces received from the other sources]
  size_t pack_cap = 0;             // doubles
  double* d_panel[2] = {nullptr, nullptr};   // gathered panel, dense, global row order (double-buffered for the look-ahead)
  size_t panel_cap[2] = {0, 0};    // doubles
  lpgp::Layout2D layout() const {
    lpgp::Layout2D l;
    l.rows.P = pr; l.rows.me = rank / pc; l.rows.nbt = (int32_t)(nb / lpgp::TILE);
    l.cols.P = pc; l.cols.me = rank % pc; l.cols.nbt = (int32_t)(nb / lpgp::TILE);
    return l;
  }
  // kernels whose dynamic-LDS attribute has been raised on THIS context's device (the attribute is
  // per device, a process-wide flag would skip it for a second context on another GPU)
  std::vector<const void*> lds_attr_done;
  // profiling
  int prof_on = 0;                 // bit k: bracket launches of kernel id k with HIP events
  int prof_open = 0;
  lpgp::ProfSlot prof[LPGP_K_COUNT];
  std::vector<lpgp::PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
};

struct lpgp_pts {
  lpgp_ctx* ctx;
  int64_t n;
  int32_t d;
  double* x;                       // device, SoA: x[dim * n_pad + i]
  int64_t n_pad;
  size_t bytes = 0;                // size of the allocation behind x
};

struct lpgp_block {
  int64_t n;                       // logical rows
  int64_t off;                     // logical offset
  int64_t poff;                    // padded offset (multiple of TILE)
  int64_t pn;                      // padded rows
};

struct lpgp_mat {
  lpgp_ctx* ctx;
  int64_t cap;                     // padded capacity of the GLOBAL matrix (multiple of TILE); single GPU: == leading dimension
  int64_t lr_cap = 0, lc_cap = 0;  // rows / columns of this rank's share (== cap on a single GPU); lr_cap is the leading dimension of a
  double* dblk = nullptr;          // multi-GPU only: the nb x nb diagonal blocks of the factor, replicated on every rank:
                                   //   block K at dblk + K * nb * nb, column-major, leading dimension nb
  double* a;                       // device lr_cap x lc_cap column-major (lower part meaningful): this rank's tiles
  double* linv;                    // device (cap/TILE) tiles of TILE x TILE: inverse of each diagonal tile of L
  double* w;                       // device 2*cap: [representer weights | residual r] (padded layout)
  std::vector<lpgp_block> blocks;  // blocks of the current VIEW (lpgp_mat_set_view): a leading run of the observation blocks
  std::vector<lpgp_block> hidden;  // blocks behind the view (appended by later conditionings; their part of the factor stays in place)
  int64_t n;                       // logical size (of the view)
  int64_t pn;                      // padded size in use (of the view)
  int64_t pn_fact;                 // padded columns factored so far (of the view)
  int64_t pn_fact_all = 0;         // the same over view + hidden blocks (meaningful while hidden is non-empty)
  int has_w;
  int has_r;                       // residual resident (lpgp_mat_set_residual)
  int* d_status = nullptr;         // device word: first non-positive pivot of the factorisations enqueued since the last lpgp_mat_check (sticky)
  int unchecked = 0;               // a factorisation was enqueued (lpgp_potrf_enqueue) and its status not read yet
  int status_known = 0;            // lpgp_potrf_predict read the status word back with its results: lpgp_mat_check needs no copy
  int status_value = 0;
  int poisoned = 0;                // a factorisation of this matrix ended undefined -- a hand-over of a resident kernel timed out (device status < 0), or a
                                   // launch failed half way through the in-place factorisation: every entry point that reads or extends the factor fails from then on
  double* r() const { return w + cap; }
};

struct lpgp_rhs {
  lpgp_ctx* ctx;
  int64_t ld;                      // padded row capacity (multiple of TILE)
  int64_t m;                       // columns
  int64_t m_pad;                   // multiple of TILE, > m: column m is spare (carries the residual through the solve)
  double* v;                       // device ld x m_pad column-major
  std::vector<char> assembled;     // per observation block: rows written by lpgp_cross_assemble (the others are cleared on first use)
};

namespace lpgp {

// every C-ABI entry point runs on its context's device whatever the calling thread's current device is
// (a negative status used to be reported ONCE -- lpgp_mat_check cleared `unchecked` first -- and the next prediction ran on a garbage factor)
#define LPGP_MAT_ALIVE(mat, fn) \
  LPGP_CHECK(!(mat)->poisoned, fn ": the factor of this matrix is undefined (an earlier factorisation timed out or failed half way); condition again from the prior")

#define LPGP_DEVICE(ctx)                                        \
  do {                                                          \
    LPGP_CHECK((ctx) != nullptr, "null context handle");        \
    LPGP_HIP(hipSetDevice((ctx)->device));                      \
  } while (0)

// raise the dynamic shared-memory limit of `fn` once per context (= per device)
inline int ensure_lds_attr(lpgp_ctx* ctx, const void* fn, size_t bytes) {
  for (const void* f : ctx->lds_attr_done)
    if (f == fn) return 0;
  LPGP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  ctx->lds_attr_done.push_back(fn);
  return 0;
}

// Wait for `stream` (dist.hip).  With an RCCL communicator of more than one rank this is NOT hipStreamSynchronize: a peer
// that failed has aborted its communicator, and a kernel of this rank that still waits for that peer's data would wait for
// ever.  The stream is polled instead, RCCL's asynchronous error state is checked every few milliseconds, and on a remote
// error -- or after LPGP_DIST_TIMEOUT_S seconds (default 600, 0: none) -- this rank aborts its own communicator, which
// terminates the pending kernels, and returns an error (the context is unusable afterwards, `dist_broken`).
int sync_stream(lpgp_ctx* ctx, hipStream_t stream);
// One tiny message between every pair of ranks and one all-reduce, checked (dist.hip; called by lpgp_dist_init while every
// rank is known to be alive): RCCL sets its connections up at first use, inside a BLOCKING host call -- later, with a peer
// gone, that call would never return; after this every exchange only enqueues kernels, which sync_stream can watch.
int dist_warm_up(lpgp_ctx* ctx);
// Link probe (dist.hip; lpgp_dist_link_probe): measured unidirectional rate of every ordered pair of ranks, of one rank
// sending to all others at once and of all ranks doing so at once, through the transport the panel exchanges use.
int dist_link_probe(lpgp_ctx* ctx, int64_t bytes, int32_t reps, double* out);

// device memory pool (api.hip)
int pool_alloc(lpgp_ctx* ctx, void** out, size_t bytes, bool* fresh);
void pool_free(lpgp_ctx* ctx, void* p, size_t bytes);

// profiling helpers: bracket launches of `kernel` on `stream`
void prof_begin(lpgp_ctx* ctx, hipStream_t stream, int kernel, double flops, double bytes);
void prof_end(lpgp_ctx* ctx, hipStream_t stream);
int prof_collect(lpgp_ctx* ctx);

// gemm.hip --------------------------------------------------------------------------------
// C(m x n) = beta*C + alpha*A*B.  All of m, n multiples of TILE, k multiple of 16.
// ta: A element (i,kk) at A[i + kk*lda] (0) or A[kk + i*lda] (1);
// tb: B element (kk,j) at B[j + kk*ldb] (0) or B[kk + j*ldb] (1).
// tri: 0 full; 1 / 2 / 3 lower-only (C(0,0) is a diagonal tile): tiles with row tile < column
//      tile are skipped (1: rank-nb trailing update, remainder
//      half; 3: its look-ahead half; 2: rank-128 update inside a panel -- same code, own kernel
//      symbol and profiling slot each).
struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc;
  int32_t mt, nt;                  // tiles in m and n
  int32_t k;
  double alpha, beta;
  int32_t tri;
  int32_t occ3 = 1;                // the launch may use the three-workgroups-per-CU kernel (the schedulers clear it while the panel chain runs beside the update: see factor_columns)
  int32_t sshift = 3;              // log2 of the super-tile edge (set by launch_gemm; legacy mapping)
  // dense tile enumeration (set by launch_gemm; map_tile_dense in gemm.hip)
  int32_t dense = 0;
  int32_t ntiles = 0, chunk = 0, nbands = 0, band = 8;
  static constexpr int MAXB = 192; // bands of 8 tile rows: 1536 tile rows = 196 608 matrix rows (> 288 GB of fp64)
  int32_t band_prefix[MAXB + 1];   // triangular shapes only: tiles before band b
  // distributed trailing update (cyc != 0, triangular shapes, dense enumeration): C is a region of this rank's
  // LOCAL tiles -- local tile row tr / column tc stand for the global tiles gr = cyc_l2g(rowc, rt0 + tr) and
  // gc = cyc_l2g(colc, ct0 + tc) -- and both operands are rows of ONE dense panel in global row order that
  // starts at global tile g0: A tile = A + (gr - g0) * 128, B tile = B + (gc - g0) * 128.  A tile is valid iff
  // gr >= gc: a staircase, enumerated band by band like the triangle (map_tile_stair in gemm.hip).
  int32_t cyc = 0;
  Cyc rowc, colc;
  int32_t rt0 = 0, ct0 = 0, g0 = 0;
  unsigned long long* stamps = nullptr;   // diagnostic builds (-DLPGP_STAMP) only
  unsigned long long* timeline = nullptr; // diagnostic builds: per-workgroup life cycle + hardware id
};
int launch_gemm(lpgp_ctx* ctx, hipStream_t stream, int ta, int tb, const GemmArgs& g, int prof_kernel);
int stair_enumerate_host(const GemmArgs& g, int32_t* out, int64_t cap);
int64_t gemm_valid_tiles(const GemmArgs& g);
// Tile solves with one step of iterative refinement (solve.hip: tile_solve_kernel); L = the 128 x 128 diagonal
// tile of the factor (leading dimension ldl, zeros above the diagonal), linv its explicit inverse (ld 128):
// X (mt*128 rows x 128, column-major ldx) <- X L^{-T} in place
// V (128 rows x nt*128 columns, column-major ldv) <- L^{-1} V in place (tile step of the forward substitution)
int launch_trsv_tile(lpgp_ctx* ctx, hipStream_t stream, double* V, int64_t ldv, const double* linv, const double* L, int64_t ldl,
                     int nt, int prof_kernel);
int launch_trsm_tile(lpgp_ctx* ctx, hipStream_t stream, double* X, int64_t ldx, const double* linv, const double* L, int64_t ldl,
                     int mt, int prof_kernel);
// the whole chain of a panel of the forward substitution in one launch (solve_panel.h: panel_solve_kernel): V (nt_rows <= 4
// tiles of rows, top row at V) <- L_KK^{-1} V; linv: the panel's tile inverses (contiguous), L: its diagonal block
int launch_trsv_panel(lpgp_ctx* ctx, hipStream_t stream, double* V, int64_t ldv, const double* linv, const double* L, int64_t ldl,
                      int nt_rows, int nt_cols, int prof_kernel);
// the same chain for rows: X (mt tiles of rows x nt_cols <= 4 tile columns) <- X L_KK^{-T}, L_KK already factored
// the same with the look-ahead update by the previous panel (4 tiles = 512 solved rows right above V) fused in front (solve4p.hip)
int launch_trsv_panel_ahead(lpgp_ctx* ctx, hipStream_t stream, double* V, int64_t ldv, const double* linv, const double* L, const double* Lprev,
                            int64_t ldl, int nt_rows, int nt_cols, int prof_kernel);
int launch_trsm_panel(lpgp_ctx* ctx, hipStream_t stream, double* X, int64_t ldx, const double* linv, const double* L, int64_t ldl,
                      int nt_cols, int mt, int prof_kernel);

// chain.hip: the whole chain of panel [p0, p0 + 4) (rows down to tile T) in one launch
int launch_panel_chain(lpgp_ctx* ctx, hipStream_t stream, lpgp_mat* mat, int p0, int T, int* d_info, bool rows_here = true, bool ahead = false);
int launch_panel_chain_rows(lpgp_ctx* ctx, hipStream_t stream, lpgp_mat* mat, int p0, int T, int* d_info);
int launch_panel_chain_v(lpgp_ctx* ctx, hipStream_t stream, lpgp_mat* mat, int p0, double* V, int64_t ldv, int64_t cols, int* d_info);
// potrf.hip -------------------------------------------------------------------------------
int debug_tile_xcc(int32_t* out8, int reset);
int launch_potrf_tile(lpgp_ctx* ctx, hipStream_t stream, double* a, int64_t lda, double* linv,
                      int* d_info, int info_base);
int potrf_blocked(lpgp_ctx* ctx, lpgp_mat* mat, int64_t t_done, int64_t T, int32_t* info);
int potrf_predict_blocked(lpgp_ctx* ctx, lpgp_mat* mat, int64_t t_done, int64_t T, double* v, int64_t ldv, int64_t m_pad);
// dist.hip: Pr x Pc block-cyclic factorisation and panel-streaming solves
int potrf_dist(lpgp_ctx* ctx, lpgp_mat* mat, int64_t t_done, int64_t T, int32_t* info);
int trsm_lower_dist(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T, double* v, int64_t ldv, int64_t m_pad);
int trsm_lower_t_dist(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T, double* v, int64_t ldv, int64_t m_pad);
int factor_to_host_dist(lpgp_ctx* ctx, lpgp_mat* mat, double* out_padded /* pn x pn column-major */);
int copy2d(hipStream_t st, double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols);
int trsm_lower_blocked(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T, double* v, int64_t ldv, int64_t m_pad);
int trsm_lower_t_blocked(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T, double* v, int64_t ldv, int64_t m_pad);
// v <- G^{-1} v; tmp: T * 128 + 2 doubles of scratch; info: device status word (INT_MIN after a timed-out hand-over of the resident form)
int solve_vec(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T, double* v, double* tmp, int* info);
int solve_vec_resident(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T, double* v, double* tmp, int* info);      // trsv.hip
int solve_vec_fwd(lpgp_ctx* ctx, hipStream_t st, lpgp_mat* mat, int64_t T, double* b, double* x);

// assemble.hip ----------------------------------------------------------------------------
// Every assembly kernel writes element (row_off + i, col_off + j) of the GLOBAL padded matrix to where `lay` keeps
// it on this rank -- or not at all if another rank owns it (single GPU: the identity layout).
int launch_assemble(lpgp_ctx* ctx, hipStream_t stream, const DevDesc& host_desc, const double* x0,
                    int64_t n0, int64_t n0_pad, const double* x1, int64_t n1, int64_t n1_pad,
                    double* out, int64_t ld, int64_t row_off, int64_t col_off, int lower_only,
                    const Layout2D& lay = Layout2D());
// several blocks with ONE descriptor (a block row of a conditioning, the rows of a cross-covariance) in one launch
struct AsmJob {
  const double* x0; int64_t n0, n0_pad;
  const double* x1; int64_t n1, n1_pad;
  int64_t row_off, col_off;
  int lower_only;
};
bool assemble_same_fast(const DevDesc& p, const DevDesc& q);
int launch_assemble_batch(lpgp_ctx* ctx, hipStream_t stream, const DevDesc& host_desc, const AsmJob* jobs, int njobs, double* out, int64_t ld,
                          const Layout2D& lay);
int launch_assemble_kron(lpgp_ctx* ctx, hipStream_t stream, const lpgp_kdesc* kd, int ngroups,
                         const double* const* F0, const int64_t* n0d, const double* const* F1, const int64_t* n1d,
                         double* work, size_t work_doubles, double* out, int64_t ld, int64_t row_off,
                         int64_t col_off, int lower_only, const Layout2D& lay);
size_t kron_work_doubles(int D, const int64_t* n0d, const int64_t* n1d);
constexpr int MV_RHS = 4;         // right-hand sides per pass of the matrix-free product (== MV_R in assemble.hip)
bool kron_fits(const lpgp_kdesc* kd, int ngroups);
// v: [r][v_stride] (0: n1_pad), out: [r][out_stride] (0: n0_pad); accumulate: out += instead of =
int launch_matvec(lpgp_ctx* ctx, hipStream_t stream, const DevDesc& host_desc, const double* x0, int64_t n0,
                  int64_t n0_pad, const double* x1, int64_t n1, int64_t n1_pad, const double* v, int nr,
                  double* part, int splits, double* out, int64_t v_stride = 0, int64_t out_stride = 0, int accumulate = 0);
int launch_add_diag(hipStream_t stream, double* a, int64_t ld, int64_t off, int64_t n, const double* v, double scalar,
                    const Layout2D& lay = Layout2D());
int launch_add_dense(hipStream_t stream, double* a, int64_t ld, int64_t off, int64_t n, const double* b,
                     const Layout2D& lay = Layout2D());

}  // namespace lpgp
