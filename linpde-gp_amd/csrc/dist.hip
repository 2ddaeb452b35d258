// Multi-GPU factorisation and solves: one process per GPU, Pr x Pc process grid, 2-D block-cyclic tiles
// (BASELINE.json north_star "Gram matrix in 2D block-cyclic tiles ... RCCL ... of panel columns
// over xGMI").
//
// STORAGE (sharded, nothing but the diagonal blocks replicated).  Rank (r, c) = r * Pc + c keeps tile (i, j) of the
// padded matrix iff (i / nbt) % Pr == r and (j / nbt) % Pc == c (blocks of nbt = nb / 128 tiles, nb = 512), as a
// dense column-major array of its tiles in increasing global order (lpgp_mat::a, leading dimension lr_cap).
// Assembly writes only owned tiles (Layout2D in the assembly kernels): zero communication, 1 / P of the work.
// Replicated on every rank: the nb x nb diagonal blocks of the factor (lpgp_mat::dblk) and all tile inverses
// (lpgp_mat::linv) -- N * nb * 8 bytes (c4: 0.27 + 0.07 GB of a 35 GB factor): they are what every triangular
// solve of a panel needs, and keeping them removes a latency-critical message from every step of every solve.
//
// ONE COMMUNICATION PRIMITIVE: gather_panel -- the rows below a factored panel (kw <= nb columns), which live
// on the Pr ranks of process column K % Pc, are made available to every rank as one dense panel in global row
// order.  Each source packs its rows (one strided copy) and pushes the SAME buffer to every other rank with
// point-to-point sends posted in one RCCL group (ncclSend / ncclRecv): direct peer-to-peer copies over the xGMI
// links of the full mesh, all links of a source busy at once, no ring.  Per panel a link carries S / Pr bytes
// (S = panel bytes), with Pc = 1 that is S / P -- the reason the default grid is P x 1 (see choose_grid).  A
// receiver unpacks the Pr pieces into the dense panel (one strided copy per piece).
//
// FACTORISATION, per panel K (right-looking): the owner of the diagonal block factors it (tile Cholesky chain on
// kw <= 4 tiles) and broadcasts block + tile inverses to everyone; the Pr ranks of the panel's process column
// solve their rows against it (refined tile solves, gemm.hip) -- the panel triangular solve runs Pr-fold
// parallel; gather_panel; every rank updates ITS tiles of the trailing matrix straight from the gathered panel:
// local tile (tr, tc) stands for global tiles (gr, gc) and reads panel rows gr and gc (cyclic tile maps in
// gemm_f64_kernel), valid iff gr >= gc -- a staircase, enumerated densely and XCD-balanced like the triangle
// (map_tile_stair).  A block append (t_done > 0) first runs the same step for every OLD panel with the solve and
// the update restricted to the new rows (the old panel is gathered from its owners' storage).
// Look-ahead: the update of the next panel's own columns and that panel's factorisation, solve and gather run
// on the panel stream while the update stream still applies the current panel to the rest.
//
// SOLVES stream the factor: right-hand sides are sharded by COLUMN over all P ranks (prediction points,
// `ConditionalGaussianProcess.predict`), every rank holds all rows of its columns, and panel after panel is
// gathered from its owners and applied locally -- forward (trsm_lower_dist) or in reverse order for L^T
// (trsm_lower_t_dist).  No right-hand-side data ever travels; the factor is streamed once per solve at S / Pr
// per link and never stored replicated.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "lpgp_internal.h"

namespace lpgp {

#define LPGP_NCCL(expr)                                                                    \
  do {                                                                                     \
    ncclResult_t _r = (expr);                                                              \
    if (_r != ncclSuccess) {                                                               \
      ::lpgp::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
      return -3;                                                                           \
    }                                                                                      \
  } while (0)

// LPGP_DIST_TRACE=1: one line on stderr around every call of the multi-GPU path that can block (bring-up aid)
static const bool g_trace = [] { const char* e = std::getenv("LPGP_DIST_TRACE"); return e && std::atoi(e) != 0; }();
#define DTRACE(ctx, ...)                                       \
  do {                                                         \
    if (g_trace) {                                             \
      std::fprintf(stderr, "[lpgp rank %d] ", (ctx)->rank);    \
      std::fprintf(stderr, __VA_ARGS__);                       \
      std::fprintf(stderr, "\n");                              \
      std::fflush(stderr);                                     \
    }                                                          \
  } while (0)

int sync_stream(lpgp_ctx* ctx, hipStream_t st) {
  DTRACE(ctx, "sync_stream: begin");
  if (!(ctx->nccl_comm && ctx->world > 1)) {
    LPGP_HIP(hipStreamSynchronize(st));
    return 0;
  }
  static const double timeout_s = [] { const char* e = std::getenv("LPGP_DIST_TIMEOUT_S"); return e ? std::atof(e) : 600.0; }();
  using clock = std::chrono::steady_clock;
  const auto t0 = clock::now();
  auto last = t0;
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) {
      DTRACE(ctx, "sync_stream: done");
      return 0;
    }
    if (q != hipErrorNotReady) {
      (void)hipGetLastError();
      set_error("waiting for the panel stream: %s", hipGetErrorString(q));
      return -1;
    }
    const auto now = clock::now();
    if (now - last > std::chrono::milliseconds(2)) {
      last = now;
      ncclComm_t comm = (ncclComm_t)ctx->nccl_comm;
      ncclResult_t aerr = ncclSuccess;
      ncclResult_t r = ncclCommGetAsyncError(comm, &aerr);
      if (r == ncclSuccess && (aerr == ncclSuccess || aerr == ncclInProgress) && ctx->nccl_comm_bulk)
        r = ncclCommGetAsyncError((ncclComm_t)ctx->nccl_comm_bulk, &aerr);
      const double waited = std::chrono::duration<double>(now - t0).count();
      const bool remote = r != ncclSuccess || (aerr != ncclSuccess && aerr != ncclInProgress);
      const bool late = timeout_s > 0.0 && waited > timeout_s;
      if (remote || late) {
        if (remote)
          set_error("a peer of the job failed: RCCL reports '%s' while this rank waited for its data (%.1f s)",
                    ncclGetErrorString(r != ncclSuccess ? r : aerr), waited);
        else
          set_error("no progress on the panel stream for %.0f s (LPGP_DIST_TIMEOUT_S): communicator aborted", waited);
        (void)ncclCommAbort(comm);           // terminates this rank's pending RCCL kernels; the peers notice the same way
        if (ctx->nccl_comm_bulk) (void)ncclCommAbort((ncclComm_t)ctx->nccl_comm_bulk);
        ctx->nccl_comm = ctx->nccl_comm_bulk = nullptr;
        ctx->dist_broken = 1;
        return -3;
      }
    }
    // spin for the first 200 us (exchanges of the panel chain are short), then back off
    if (now - t0 < std::chrono::microseconds(200)) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}

static inline GemmArgs mk(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int mt, int nt,
                          int k, double alpha, double beta, int tri) {
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.mt = mt; g.nt = nt; g.k = k; g.alpha = alpha; g.beta = beta;
  g.tri = tri;
  return g;
}

// ---- broadcast of a set of device buffers, each from its own root to every rank ---------------------------------
struct Piece {
  int root;
  double* buf;           // source on the root, destination on every rank of `dst`
  size_t count;          // doubles
  uint64_t dst = ~0ull;  // ranks that receive the piece (bit r = rank r; the root's own bit is ignored).  Default: everyone.
  bool to(int rank) const { return (dst >> rank) & 1ull; }
};

static int bcast_pieces(lpgp_ctx* ctx, hipStream_t st, const std::vector<Piece>& pieces, bool bulk = false) {
  if (ctx->world <= 1) return 0;
  double sent = 0.0, recv = 0.0;
  for (const auto& p : pieces) {
    if (p.count == 0) continue;
    if (p.root == ctx->rank) {
      for (int peer = 0; peer < ctx->world; ++peer)
        if (peer != ctx->rank && p.to(peer)) sent += 8.0 * (double)p.count;
    } else if (p.to(ctx->rank)) {
      recv += 8.0 * (double)p.count;
    }
  }
  ctx->comm_bytes_sent += sent;
  ctx->comm_bytes_recv += recv;
  prof_begin(ctx, st, LPGP_K_COMM, 0.0, sent + recv);
  if (ctx->ipc()) {
    // direct-peer transport.  The pieces of one exchange are laid out back to back in every rank's window (same
    // offsets everywhere); an exchange that does not fit is cut into rounds.  Per round:
    //   wait       this rank's copy-out of the PREVIOUS round of this window has completed (an event, not a drain)
    //   barrier A  ... on every rank: the window may be written again
    //   push       each root copies its pieces into the window of every destination, device to device, and waits for them
    //   barrier B  all pushes have landed
    //   copy out   window -> destination buffers, asynchronously on `st`; its completion is the window's event
    // Round 4: every rank's window is used as TWO halves, one per exchange stream -- the exchanges of the panel stream
    // (diagonal blocks, gathers, the head of a split gather) in the lower half, the bulk exchanges of the exchange stream
    // (the tail of a split gather) in the upper one -- so that an exchange waits for ITS OWN previous copy-out only.
    // (Until round 3 the one window was shared by both streams and every exchange drained the panel stream AND the
    // exchange stream before a peer could push again: the look-ahead could hide none of it.)  The barriers still travel
    // over the control plane: this transport is the fallback for boxes where RCCL cannot be brought up.
    auto barrier = [&]() -> int {
      int v = 0;
      LPGP_CHECK(ctx->host_xfer(ctx->host_xfer_user, 1, &v, (int64_t)sizeof(int), 0) == 0, "ipc transport: barrier failed");
      return 0;
    };
    const int half = (bulk && ctx->s_comm && st == ctx->s_comm) ? 1 : 0;
    const size_t wdoubles = ctx->ipc_window_doubles / 2, woff = (size_t)half * wdoubles;
    if (!ctx->ev_ipc[half]) LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_ipc[half], hipEventDisableTiming));
    auto window_free = [&]() -> int {
      if (ctx->ipc_copyout_pending[half]) LPGP_HIP(hipEventSynchronize(ctx->ev_ipc[half]));
      ctx->ipc_copyout_pending[half] = 0;
      return 0;
    };
    auto copied_out = [&]() -> int {
      LPGP_HIP(hipEventRecord(ctx->ev_ipc[half], st));
      ctx->ipc_copyout_pending[half] = 1;
      return 0;
    };
    size_t i = 0;
    while (i < pieces.size()) {
      // one round: pieces [i, j) (a piece larger than the window travels in slices)
      size_t used = 0, j = i;
      std::vector<size_t> offs;
      while (j < pieces.size() && used + pieces[j].count <= wdoubles) {
        offs.push_back(used);
        used += pieces[j].count;
        ++j;
      }
      if (j == i) {
        // a single piece exceeds the window: slices of the window's size
        const Piece& p = pieces[i];
        for (size_t o = 0; o < p.count; o += wdoubles) {
          const size_t n = std::min(wdoubles, p.count - o);
          LPGP_TRY(window_free());
          LPGP_TRY(barrier());
          if (p.root == ctx->rank) {
            for (int peer = 0; peer < ctx->world; ++peer)
              if (peer != ctx->rank && p.to(peer))
                LPGP_HIP(hipMemcpyAsync(ctx->ipc_peer[peer] + woff, p.buf + o, n * sizeof(double), hipMemcpyDeviceToDevice, st));
            LPGP_HIP(hipStreamSynchronize(st));
          }
          LPGP_TRY(barrier());
          if (p.root != ctx->rank && p.to(ctx->rank))
            LPGP_HIP(hipMemcpyAsync(p.buf + o, ctx->ipc_window + woff, n * sizeof(double), hipMemcpyDeviceToDevice, st));
          LPGP_TRY(copied_out());
        }
        ++i;
        continue;
      }
      LPGP_TRY(window_free());
      LPGP_TRY(barrier());
      bool pushed = false;
      for (size_t q = i; q < j; ++q) {
        const Piece& p = pieces[q];
        if (p.count == 0 || p.root != ctx->rank) continue;
        for (int peer = 0; peer < ctx->world; ++peer)
          if (peer != ctx->rank && p.to(peer))
            LPGP_HIP(hipMemcpyAsync(ctx->ipc_peer[peer] + woff + offs[q - i], p.buf, p.count * sizeof(double), hipMemcpyDeviceToDevice, st));
        pushed = true;
      }
      if (pushed) LPGP_HIP(hipStreamSynchronize(st));
      LPGP_TRY(barrier());
      for (size_t q = i; q < j; ++q) {
        const Piece& p = pieces[q];
        if (p.count == 0 || p.root == ctx->rank || !p.to(ctx->rank)) continue;
        LPGP_HIP(hipMemcpyAsync(p.buf, ctx->ipc_window + woff + offs[q - i], p.count * sizeof(double), hipMemcpyDeviceToDevice, st));
      }
      LPGP_TRY(copied_out());
      i = j;
    }
  } else if (ctx->host_xfer) {
    // bring-up / test transport: every piece staged through the host and broadcast by the caller's exchange
    std::vector<double> stage;
    for (const auto& p : pieces) {
      if (p.count == 0) continue;
      if (stage.size() < p.count) stage.resize(p.count);
      if (p.root == ctx->rank) LPGP_HIP(hipMemcpyAsync(stage.data(), p.buf, p.count * sizeof(double), hipMemcpyDeviceToHost, st));
      LPGP_HIP(hipStreamSynchronize(st));
      LPGP_CHECK(ctx->host_xfer(ctx->host_xfer_user, 0, stage.data(), (int64_t)(p.count * sizeof(double)), p.root) == 0,
                 "host exchange: broadcast from rank %d failed", p.root);
      if (p.root != ctx->rank && p.to(ctx->rank)) {        // (the caller's exchange is a broadcast; a rank outside `dst` drops it)
        LPGP_HIP(hipMemcpyAsync(p.buf, stage.data(), p.count * sizeof(double), hipMemcpyHostToDevice, st));
        LPGP_HIP(hipStreamSynchronize(st));
      }
    }
  } else {
    ncclComm_t comm = (ncclComm_t)((bulk && ctx->nccl_comm_bulk) ? ctx->nccl_comm_bulk : ctx->nccl_comm);
    LPGP_CHECK(comm != nullptr && ctx->nccl_comm != nullptr, "panel exchange: no communicator (aborted after an earlier failure?)");
    // LPGP_DIST_COLLECTIVE=bcast: one ncclBroadcast per piece instead of the point-to-point group (a fallback to compare
    // with on the 8-GPU node, which the builder has no access to; RCCL then picks its own ring / tree)
    const bool use_bcast = ctx->dist_bcast != 0;
    DTRACE(ctx, "exchange: group of %zu pieces", pieces.size());
    LPGP_NCCL(ncclGroupStart());
    for (const auto& p : pieces) {
      if (p.count == 0) continue;
      if (use_bcast) {
        LPGP_NCCL(ncclBroadcast(p.buf, p.buf, p.count, ncclDouble, p.root, comm, st));
      } else if (p.root == ctx->rank) {
        for (int peer = 0; peer < ctx->world; ++peer)
          if (peer != ctx->rank && p.to(peer)) LPGP_NCCL(ncclSend(p.buf, p.count, ncclDouble, peer, comm, st));
      } else if (p.to(ctx->rank)) {
        LPGP_NCCL(ncclRecv(p.buf, p.count, ncclDouble, p.root, comm, st));
      }
    }
    LPGP_NCCL(ncclGroupEnd());
    DTRACE(ctx, "exchange: group enqueued");
  }
  prof_end(ctx, st);
  return 0;
}

static int allreduce_max_int(lpgp_ctx* ctx, hipStream_t st, int* h_value) {
  if (ctx->world <= 1) return 0;
  if (ctx->host_xfer) {     // host and direct-peer transports: the control-plane exchange
    LPGP_CHECK(ctx->host_xfer(ctx->host_xfer_user, 1, h_value, (int64_t)sizeof(int), 0) == 0, "host exchange: all-reduce failed");
    return 0;
  }
  // through PINNED host memory: a copy to or from pageable memory is synchronous -- it would wait, unwatched, behind a
  // collective whose peer is gone; like this only sync_stream waits, and it watches RCCL's error state
  int* d = ctx->d_info;
  int* hp = ctx->h_info_pinned;
  *hp = *h_value;
  LPGP_HIP(hipMemcpyAsync(d, hp, sizeof(int), hipMemcpyHostToDevice, st));
  DTRACE(ctx, "all-reduce: enqueue");
  LPGP_NCCL(ncclAllReduce(d, d, 1, ncclInt, ncclMax, (ncclComm_t)ctx->nccl_comm, st));
  LPGP_HIP(hipMemcpyAsync(hp, d, sizeof(int), hipMemcpyDeviceToHost, st));
  LPGP_TRY(sync_stream(ctx, st));
  *h_value = *hp;
  return 0;
}

int dist_warm_up(lpgp_ctx* ctx) {
  if (ctx->world <= 1 || !ctx->nccl_comm) return 0;
  hipStream_t st = ctx->s_main;
  const int W = ctx->world;
  {
    // ncclCommSplit is checked locally only: agree on its outcome (max over the ranks of "I have none") on the MAIN
    // communicator, and drop the bulk communicator everywhere unless every rank got one -- ranks that disagree would
    // post their bulk exchanges on different communicators and wait for each other until the watchdog fires
    int none = ctx->nccl_comm_bulk ? 0 : 1;
    LPGP_TRY(allreduce_max_int(ctx, st, &none));
    if (none && ctx->nccl_comm_bulk) {
      (void)ncclCommDestroy((ncclComm_t)ctx->nccl_comm_bulk);
      ctx->nccl_comm_bulk = nullptr;
    }
  }
  double* d = nullptr;
  LPGP_HIP(hipMalloc(&d, (size_t)W * sizeof(double)));
  std::vector<double> h((size_t)W, -1.0);
  h[(size_t)ctx->rank] = 1000.0 + ctx->rank;
  int rc = 0;
  do {
    if (hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) { rc = -1; break; }
    std::vector<Piece> pieces;
    for (int r = 0; r < W; ++r) pieces.push_back({r, d + r, 1});
    if ((rc = bcast_pieces(ctx, st, pieces)) != 0) break;
    if (ctx->nccl_comm_bulk && (rc = bcast_pieces(ctx, st, pieces, true)) != 0) break;
    if (hipMemcpyAsync(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) { rc = -1; break; }
    if ((rc = sync_stream(ctx, st)) != 0) break;
    for (int r = 0; r < W; ++r)
      if (h[(size_t)r] != 1000.0 + r) {
        set_error("communicator warm-up: rank %d received %g from rank %d (expected %g)", ctx->rank, h[(size_t)r], r, 1000.0 + r);
        rc = -3;
      }
    if (rc != 0) break;
    int v = ctx->rank;
    if ((rc = allreduce_max_int(ctx, st, &v)) != 0) break;
    if (v != W - 1) {
      set_error("communicator warm-up: all-reduce(max) of the ranks gave %d, expected %d", v, W - 1);
      rc = -3;
    }
  } while (0);
  (void)hipFree(d);
  return rc;
}

// ---- link probe ---------------------------------------------------------------------------------------------------
// What the panel exchanges can expect from the fabric, measured through the very calls they use (RCCL: ncclSend / ncclRecv
// groups on the panel stream; direct-peer transport: hipMemcpyAsync into the peer's IPC-mapped window), timed with HIP
// events on that stream.  out (W*W + W + 1 doubles, GB/s, 0 = not measured on this rank):
//   out[s*W + d]   s -> d alone                     (RCCL: measured by the receiver d; IPC: by the sender s)
//   out[W*W + s]   s -> every peer at once: rate of ONE of its links (RCCL: inbound at this rank; IPC: at the sender, per peer)
//   out[W*W + W]   every rank -> every peer at once: total inbound rate of this rank (the pattern of a P x 1 panel gather)
//   out[W*W + W + 1]  bytes per message actually transferred
// Collective; every rank passes the same bytes / reps.  bench.py gathers the rows and prints the matrix (config.link_probe).
int dist_link_probe(lpgp_ctx* ctx, int64_t bytes, int32_t reps, double* out) {
  const int W = ctx->world, me = ctx->rank;
  for (int i = 0; i < W * W + W + 2; ++i) out[i] = 0.0;
  if (W <= 1 || (ctx->host_xfer && !ctx->ipc())) return 0;       // host-staged bring-up transport: nothing to measure
  hipStream_t st = ctx->s_main;
  size_t count = (size_t)bytes / sizeof(double);
  if (ctx->ipc()) count = std::min(count, ctx->ipc_window_doubles / 2 / (size_t)W);      // (lower half: the exchanges of the panel stream)
  out[W * W + W + 1] = (double)(count * sizeof(double));      // bytes per message actually moved (the IPC window may cap the request)
  void *ps = nullptr, *pr = nullptr;
  const size_t sb = count * sizeof(double), rb = sb * (size_t)(W - 1);
  if (pool_alloc(ctx, &ps, sb, nullptr) != 0) return -1;
  if (pool_alloc(ctx, &pr, rb, nullptr) != 0) { pool_free(ctx, ps, sb); return -1; }
  double* dsend = (double*)ps;
  double* drecv = (double*)pr;
  struct Release { lpgp_ctx* c; void *a, *b; size_t sa, sb_; ~Release() { pool_free(c, a, sa); pool_free(c, b, sb_); } } release{ctx, ps, pr, sb, rb};
  LPGP_HIP(hipMemsetAsync(dsend, 0, sb, st));
  hipEvent_t e0, e1;
  LPGP_HIP(hipEventCreate(&e0));
  LPGP_HIP(hipEventCreate(&e1));
  struct Ev { hipEvent_t a, b; ~Ev() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evs{e0, e1};
  auto barrier = [&]() -> int {
    int v = 0;
    return allreduce_max_int(ctx, st, &v);
  };
  // one pattern: rank s sends `count` doubles to every d with sends(s, d); returns the seconds per repetition seen by this rank
  auto run = [&](auto&& sends, double* seconds) -> int {
    *seconds = 0.0;
    bool involved = false;
    for (int s_ = 0; s_ < W && !involved; ++s_)
      for (int d_ = 0; d_ < W; ++d_)
        if (s_ != d_ && sends(s_, d_) && (s_ == me || d_ == me)) { involved = true; break; }
    double ipc_ms = 0.0;
    for (int r = 0; r <= reps; ++r) {          // r == 0: untimed (connection set-up, both ends in step)
      if (r == 1 && involved && !ctx->ipc()) LPGP_HIP(hipEventRecord(e0, st));
      if (ctx->ipc()) {
        // direct-peer transport: the barrier between repetitions (a control-plane round trip) stays OUTSIDE the timed
        // interval -- only the copies are timed, repetition by repetition (the rates used to include it)
        LPGP_HIP(hipStreamSynchronize(st));
        LPGP_TRY(barrier());
        if (involved && r >= 1) LPGP_HIP(hipEventRecord(e0, st));
        for (int d_ = 0; d_ < W; ++d_)
          if (d_ != me && sends(me, d_))
            LPGP_HIP(hipMemcpyAsync(ctx->ipc_peer[d_] + (size_t)me * count, dsend, sb, hipMemcpyDeviceToDevice, st));
        if (involved && r >= 1) {
          LPGP_HIP(hipEventRecord(e1, st));
          LPGP_HIP(hipStreamSynchronize(st));
          float ms = 0.f;
          LPGP_HIP(hipEventElapsedTime(&ms, e0, e1));
          ipc_ms += (double)ms;
        }
      } else {
        ncclComm_t comm = (ncclComm_t)ctx->nccl_comm;
        LPGP_CHECK(comm != nullptr, "link probe: no communicator");
        if (involved) {
          LPGP_NCCL(ncclGroupStart());
          int slot = 0;
          for (int p = 0; p < W; ++p) {
            if (p == me) continue;
            if (sends(me, p)) LPGP_NCCL(ncclSend(dsend, count, ncclDouble, p, comm, st));
            if (sends(p, me)) LPGP_NCCL(ncclRecv(drecv + (size_t)slot * count, count, ncclDouble, p, comm, st));
            ++slot;
          }
          LPGP_NCCL(ncclGroupEnd());
        }
      }
    }
    if (involved && ctx->ipc()) {
      *seconds = ipc_ms * 1e-3 / reps;
    } else if (involved) {
      LPGP_HIP(hipEventRecord(e1, st));
      LPGP_TRY(sync_stream(ctx, st));
      float ms = 0.f;
      LPGP_HIP(hipEventElapsedTime(&ms, e0, e1));
      *seconds = (double)ms * 1e-3 / reps;
    }
    if (ctx->ipc()) LPGP_TRY(barrier());
    return 0;
  };
  const double gb = (double)sb * 1e-9;
  double sec = 0.0;
  for (int s_ = 0; s_ < W; ++s_)
    for (int d_ = 0; d_ < W; ++d_) {
      if (s_ == d_) continue;
      LPGP_TRY(run([&](int a, int b) { return a == s_ && b == d_; }, &sec));
      const bool mine = ctx->ipc() ? me == s_ : me == d_;
      if (mine && sec > 0.0) out[s_ * W + d_] = gb / sec;
    }
  LPGP_TRY(barrier());
  for (int s_ = 0; s_ < W; ++s_) {
    LPGP_TRY(run([&](int a, int) { return a == s_; }, &sec));
    const bool mine = ctx->ipc() ? me == s_ : me != s_;
    if (mine && sec > 0.0) out[W * W + s_] = gb / sec;
    LPGP_TRY(barrier());
  }
  LPGP_TRY(run([&](int, int) { return true; }, &sec));
  if (sec > 0.0) out[W * W + W] = gb * (W - 1) / sec;
  LPGP_TRY(barrier());
  return 0;
}

// ---- panel gather -------------------------------------------------------------------------------------------------
// rows of `src` (ntiles tiles of 128 rows, the source member's tiles l0, l0 + 1, ... in ITS local order) -> rows of
// the dense panel in global order: local tile l0 + i of member `c.me` is global tile cyc_l2g(c, l0 + i)
__global__ __launch_bounds__(256) void unpack_piece_kernel(double* __restrict__ dst, int64_t ldd, const double* __restrict__ src,
                                                            int64_t lds, Cyc c, int l0, int g_lo, int64_t rows) {
  const int64_t col = blockIdx.y;
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (r >= rows) return;
  const int i = (int)(r / TILE);
  const int64_t gr = (int64_t)(cyc_l2g(c, l0 + i) - g_lo) * TILE + (r - (int64_t)i * TILE);
  *reinterpret_cast<double2*>(dst + col * ldd + gr) = *reinterpret_cast<const double2*>(src + col * lds + r);
}

// the inverse on the source: tile i of the piece (class `cq`, its local tiles l0, l0 + 1, ...) is global tile cyc_l2g(cq, l0 + i),
// which the source member `cr` keeps as its local tile cyc_before(cr, that)
__global__ __launch_bounds__(256) void pack_piece_kernel(double* __restrict__ dst, int64_t ldd, const double* __restrict__ src,
                                                          int64_t lds, Cyc cq, Cyc cr, int l0, int64_t rows) {
  const int64_t col = blockIdx.y;
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (r >= rows) return;
  const int i = (int)(r / TILE);
  const int64_t sr = (int64_t)cyc_before(cr, cyc_l2g(cq, l0 + i)) * TILE + (r - (int64_t)i * TILE);
  *reinterpret_cast<double2*>(dst + col * ldd + r) = *reinterpret_cast<const double2*>(src + col * lds + sr);
}
// debug aid (LPGP_DIST_POISON=1, tests): the panel buffer is filled with NaN before a gather, so that an update that read a
// row this rank did not receive would poison the factor
__global__ __launch_bounds__(256) void poison_kernel(double* __restrict__ p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = __builtin_nan("");
}

static int ensure_buf(lpgp_ctx* ctx, double** p, size_t* cap, size_t doubles) {
  if (doubles <= *cap) return 0;
  DTRACE(ctx, "ensure_buf: %zu -> %zu doubles", *cap, doubles);
  // hipFree waits for the device: drain the panel stream first, WATCHED (an exchange with a dead peer may be pending on it)
  if (*p) LPGP_TRY(sync_stream(ctx, ctx->s_main));
  if (*p && ctx->s_comm) LPGP_TRY(sync_stream(ctx, ctx->s_comm));
  if (*p) LPGP_HIP(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  LPGP_HIP(hipMalloc(p, doubles * sizeof(double)));
  *cap = doubles;
  return 0;
}

struct Grid {
  Cyc R, C;              // this rank's row / column membership
  int Pr, Pc, my_r, my_c, nbt;
  Cyc Rof(int r) const { Cyc c = R; c.me = r; return c; }
};
static Grid grid_of(const lpgp_ctx* ctx) {
  Grid g;
  const Layout2D l = ctx->layout();
  g.R = l.rows; g.C = l.cols;
  g.Pr = ctx->pr; g.Pc = ctx->pc; g.my_r = l.rows.me; g.my_c = l.cols.me; g.nbt = l.rows.nbt;
  return g;
}

// Make rows [g_a, g_b) (global tiles) of tile columns [c0, c1) of the (partly) factored matrix available inside a dense
// column-major panel `out` whose first row is global tile g_lo (leading dimension (T - g_lo) * 128), global row order.  The
// columns live on process column pcK = (c0 / nbt) % Pc.  Collective; everything is enqueued on `st`.
// with_self = false: this rank's own rows are in `out` already (self_unpack) -- it still sends them.  bulk: the second
// communicator and its own pack buffer (the exchange runs beside the small exchanges of the panel chain).
// scoped = false: every rank receives every row (the streamed solves, the collected factor: a rank applies the whole
// panel to its columns).  scoped = true (round 4; the trailing update of the factorisation on Pr, Pc > 1 grids): a rank
// receives only the rows its update READS -- its tiles (gr, gc) need panel rows gr (row class my_r) and gc (column class
// my_c) -- so a block of rows with index B goes to process row B % Pr and to process column B % Pc only: the pieces are
// the classes q = B mod lcm(Pr, Pc), piece q travels from rank (q % Pr, pcK) to the ranks (r, c) with r == q % Pr or
// c == q % Pc, and a rank takes in ~ S (1/Pr + 1/Pc) of a panel of S bytes instead of S (2 x 4: 3/4).
static int gather_rows(lpgp_ctx* ctx, hipStream_t st, lpgp_mat* mat, const Grid& G, int T, int c0, int c1, int g_lo, int g_a, int g_b,
                       bool with_self, bool bulk, double* out, bool scoped = false) {
  if (g_b <= g_a || T <= g_lo) return 0;
  const int64_t ldo = (int64_t)(T - g_lo) * TILE, cols = (int64_t)(c1 - c0) * TILE;
  const int pcK = (c0 / G.nbt) % G.Pc;
  const int64_t ld = mat->lr_cap;
  if (!(G.Pr > 1 && G.Pc > 1)) scoped = false;                    // (Pr == 1: every rank holds whole tile rows; Pc == 1: whole columns)
  int Lc = G.Pr;
  if (scoped) {
    int a = G.Pr, b = G.Pc;
    while (b) { const int t = a % b; a = b; b = t; }
    Lc = G.Pr / a * G.Pc;
  }
  static const bool poison = [] { const char* e = std::getenv("LPGP_DIST_POISON"); return e && std::atoi(e) != 0; }();
  if (poison && with_self && g_a == g_lo && g_b == T) {
    const size_t n = (size_t)ldo * (size_t)cols;
    hipLaunchKernelGGL(poison_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n);
  }
  // piece of class q: its tiles [l0_q, l0_q + nt_q) in the class's own enumeration
  auto class_of = [&](int q) { Cyc c = G.R; c.P = Lc; c.me = q; return c; };
  std::vector<int> l0(Lc), nt(Lc);
  size_t total = 0;
  for (int q = 0; q < Lc; ++q) {
    const Cyc cq = class_of(q);
    l0[q] = cyc_before(cq, g_a);
    nt[q] = cyc_before(cq, g_b) - l0[q];
    total += (size_t)nt[q] * TILE * cols;
  }
  double** pack = bulk ? &ctx->d_pack_bulk : &ctx->d_pack;
  LPGP_TRY(ensure_buf(ctx, pack, bulk ? &ctx->pack_bulk_cap : &ctx->pack_cap, total));
  std::vector<Piece> pieces;
  std::vector<double*> pbuf(Lc);
  std::vector<char> mine(Lc, 1);                                  // does this rank hold piece q after the exchange?
  size_t off = 0;
  for (int q = 0; q < Lc; ++q) {
    pbuf[q] = *pack + off;
    off += (size_t)nt[q] * TILE * cols;
    if (nt[q] == 0) continue;
    const int r = q % G.Pr, root = r * G.Pc + pcK;
    uint64_t dst = ~0ull;
    if (scoped) {
      dst = 0;
      for (int rr = 0; rr < G.Pr; ++rr)
        for (int cc = 0; cc < G.Pc; ++cc)
          if (rr == r || cc == q % G.Pc) dst |= 1ull << (rr * G.Pc + cc);
      mine[q] = (dst >> ctx->rank) & 1ull;
    }
    if (root == ctx->rank && ctx->world > 1) {
      // own rows, packed for the peers (and unpacked from there into this rank's own panel below)
      const double* src = mat->a + (int64_t)cyc_before(G.C, c0) * TILE * ld;
      const int64_t rows = (int64_t)nt[q] * TILE;
      hipLaunchKernelGGL(pack_piece_kernel, dim3((unsigned)((rows / 2 + 255) / 256), (unsigned)cols), dim3(256), 0, st, pbuf[q], rows, src,
                         ld, class_of(q), G.Rof(r), l0[q], rows);
    }
    Piece pc{root, pbuf[q], (size_t)nt[q] * TILE * cols};
    pc.dst = dst;
    pieces.push_back(pc);
  }
  LPGP_HIP(hipGetLastError());
  LPGP_TRY(bcast_pieces(ctx, st, pieces, bulk));
  for (int q = 0; q < Lc; ++q) {
    if (nt[q] == 0) continue;
    const int r = q % G.Pr, root = r * G.Pc + pcK;
    if (root == ctx->rank && !with_self) continue;
    if (root != ctx->rank && !mine[q]) continue;
    const double* src = pbuf[q];
    int64_t lds = (int64_t)nt[q] * TILE;
    if (root == ctx->rank && ctx->world <= 1) {
      src = mat->a + (int64_t)l0[q] * TILE + (int64_t)cyc_before(G.C, c0) * TILE * ld;      // (one rank: class == member, contiguous)
      lds = ld;
    }
    const int64_t rows = (int64_t)nt[q] * TILE;
    hipLaunchKernelGGL(unpack_piece_kernel, dim3((unsigned)((rows / 2 + 255) / 256), (unsigned)cols), dim3(256), 0, st, out, ldo, src,
                       lds, class_of(q), l0[q], g_lo, rows);
  }
  LPGP_HIP(hipGetLastError());
  return 0;
}

static int gather_panel(lpgp_ctx* ctx, hipStream_t st, lpgp_mat* mat, const Grid& G, int T, int c0, int c1, int g_lo, double* out,
                        bool scoped = false) {
  return gather_rows(ctx, st, mat, G, T, c0, c1, g_lo, g_lo, T, true, false, out, scoped);
}

// This rank's OWN rows [g_lo, T) of tile columns [c0, c1) (it is a member of their process column) into the dense panel:
// a local copy, no communication
static int self_unpack(lpgp_ctx* ctx, hipStream_t st, lpgp_mat* mat, const Grid& G, int T, int c0, int c1, int g_lo, double* out) {
  (void)ctx;
  const int l0 = cyc_before(G.R, g_lo), nt = cyc_before(G.R, T) - l0;
  if (nt <= 0) return 0;
  const int64_t ld = mat->lr_cap, rows = (int64_t)nt * TILE, cols = (int64_t)(c1 - c0) * TILE;
  const double* src = mat->a + (int64_t)l0 * TILE + (int64_t)cyc_before(G.C, c0) * TILE * ld;
  hipLaunchKernelGGL(unpack_piece_kernel, dim3((unsigned)((rows / 2 + 255) / 256), (unsigned)cols), dim3(256), 0, st, out,
                     (int64_t)(T - g_lo) * TILE, src, ld, G.R, l0, g_lo, rows);
  LPGP_HIP(hipGetLastError());
  return 0;
}

// tile (it, jt) of diagonal block K as replicated in mat->dblk (leading dimension nb)
static inline double* dblk_tile(const lpgp_mat* mat, const Grid& G, int K, int it, int jt) {
  const int64_t nb = (int64_t)G.nbt * TILE;
  return mat->dblk + (int64_t)K * nb * nb + (int64_t)it * TILE + (int64_t)jt * TILE * nb;
}

// Rows [rl0, rl1) (LOCAL tiles of this rank, which is a member of the panel's process column) of tile columns
// [c0, c1) <- X L_KK^{-T}: kw tile steps {refined tile solve; rank-128 update of the panel columns to the right}
static int panel_rows_solve(lpgp_ctx* ctx, hipStream_t st, lpgp_mat* mat, const Grid& G, int c0, int c1, int rl0, int rl1) {
  const int mt = rl1 - rl0;
  if (mt <= 0) return 0;
  const int64_t ld = mat->lr_cap, nb = (int64_t)G.nbt * TILE;
  const int K = c0 / G.nbt, b0 = c0 - K * G.nbt;
  double* X = mat->a + (int64_t)rl0 * TILE + (int64_t)cyc_before(G.C, c0) * TILE * ld;
  if (ctx->fused_solve && c1 - c0 <= 4)        // the diagonal block is final here: the whole chain of the panel in one launch
    return launch_trsm_panel(ctx, st, X, ld, mat->linv + (int64_t)c0 * TILE * TILE, dblk_tile(mat, G, K, b0, b0), nb, c1 - c0, mt,
                             LPGP_K_PANEL);
  for (int j = 0; j < c1 - c0; ++j) {
    double* Xj = X + (int64_t)j * TILE * ld;
    LPGP_TRY(launch_trsm_tile(ctx, st, Xj, ld, mat->linv + (int64_t)(c0 + j) * TILE * TILE, dblk_tile(mat, G, K, b0 + j, b0 + j), nb, mt,
                              LPGP_K_TRSM));
    if (j + 1 < c1 - c0)
      LPGP_TRY(launch_gemm(ctx, st, 0, 0,
                           mk(Xj, ld, dblk_tile(mat, G, K, b0 + j + 1, b0 + j), nb, Xj + (int64_t)TILE * ld, ld, mt, c1 - c0 - j - 1, TILE,
                              -1.0, 1.0, 0),
                           LPGP_K_GEMM));
  }
  return 0;
}

// The update of this rank's tiles by a gathered panel: local tile rows [rt0, LTr) x local tile columns [ct0, ct1),
// valid where global row tile >= global column tile; `panel` holds global tile rows g0 ... (leading dimension ldp)
static GemmArgs update_args(lpgp_mat* mat, const Grid& G, const double* panel, int64_t ldp, int g0, int K128, int rt0, int rt1, int ct0,
                            int ct1, int tri) {
  const int64_t ld = mat->lr_cap;
  GemmArgs g = mk(panel, ldp, panel, ldp, mat->a + (int64_t)rt0 * TILE + (int64_t)ct0 * TILE * ld, ld, rt1 - rt0, ct1 - ct0, K128, -1.0, 1.0,
                  tri);
  g.cyc = 1;
  g.rowc = G.R; g.colc = G.C;
  g.rt0 = rt0; g.ct0 = ct0; g.g0 = g0;
  return g;
}
static int update_local(lpgp_ctx* ctx, hipStream_t st, lpgp_mat* mat, const Grid& G, const double* panel, int64_t ldp, int g0, int K128,
                        int rt0, int rt1, int ct0, int ct1, int prof) {
  if (rt1 <= rt0 || ct1 <= ct0) return 0;
  // (the look-ahead half keeps its own kernel symbol and profiling slot, as on a single GPU)
  return launch_gemm(ctx, st, 0, 0, update_args(mat, G, panel, ldp, g0, K128, rt0, rt1, ct0, ct1, prof == LPGP_K_SYRK_AHEAD ? 3 : 1), prof);
}

// Factor the diagonal block of panel [c0, c1) (owner only, in its local storage), copy it into the replicated
// dblk / keep its tile inverses in linv; then broadcast both to every rank.
static int factor_diag_block(lpgp_ctx* ctx, hipStream_t st, lpgp_mat* mat, const Grid& G, int T, int c0, int c1) {
  const int K = c0 / G.nbt, b0 = c0 - K * G.nbt, kw = c1 - c0;
  const int64_t ld = mat->lr_cap, nb = (int64_t)G.nbt * TILE;
  const int owner = ((K % G.Pr) * G.Pc) + (K % G.Pc);
  if (owner == ctx->rank) {
    double* D = mat->a + (int64_t)cyc_before(G.R, c0) * TILE + (int64_t)cyc_before(G.C, c0) * TILE * ld;
    for (int j = 0; j < kw; ++j) {
      double* dj = D + (int64_t)j * TILE * (ld + 1);
      double* linv = mat->linv + (int64_t)(c0 + j) * TILE * TILE;
      LPGP_TRY(launch_potrf_tile(ctx, st, dj, ld, linv, ctx->d_info, (c0 + j) * TILE));
      if (j + 1 < kw) {
        double* X = dj + TILE;
        LPGP_TRY(launch_trsm_tile(ctx, st, X, ld, linv, dj, ld, kw - j - 1, LPGP_K_TRSM));
        LPGP_TRY(launch_gemm(ctx, st, 0, 0, mk(X, ld, X, ld, dj + (int64_t)TILE * (ld + 1), ld, kw - j - 1, kw - j - 1, TILE, -1.0, 1.0, 2),
                             LPGP_K_SYRK_PANEL));
      }
    }
    LPGP_TRY(copy2d(st, dblk_tile(mat, G, K, b0, b0), nb, D, ld, (int64_t)kw * TILE, (int64_t)kw * TILE));
    // a panel that starts inside its block (block append at a boundary that is not a multiple of nb): the block's
    // rows of this panel x the block's OLD columns were solved by the append phase (they are rows below an old
    // panel) and belong to the replicated diagonal block as well
    if (b0 > 0)
      LPGP_TRY(copy2d(st, dblk_tile(mat, G, K, b0, 0), nb, D - (int64_t)b0 * TILE * ld, ld, (int64_t)kw * TILE, (int64_t)b0 * TILE));
  }
  if (ctx->world > 1) {
    // the whole block K (an earlier, partial panel of the same block is identical everywhere) and the tile
    // inverses of the block's tile columns that exist
    const int t0 = K * G.nbt, t1 = std::min(T, (K + 1) * G.nbt);
    std::vector<Piece> pieces;
    pieces.push_back({owner, mat->dblk + (int64_t)K * nb * nb, (size_t)(nb * nb)});
    pieces.push_back({owner, mat->linv + (int64_t)t0 * TILE * TILE, (size_t)(t1 - t0) * TILE * TILE});
    LPGP_TRY(bcast_pieces(ctx, st, pieces));
  }
  (void)T;
  return 0;
}

static int ensure_panel(lpgp_ctx* ctx, int which, size_t doubles) { return ensure_buf(ctx, &ctx->d_panel[which], &ctx->panel_cap[which], doubles); }

// Factor tile columns [t_done, T) of the distributed matrix; columns [0, t_done) already hold L.
int potrf_dist(lpgp_ctx* ctx, lpgp_mat* mat, int64_t t_done64, int64_t T64, int32_t* info) {
  const int T = (int)T64, t_done = (int)t_done64;
  const Grid G = grid_of(ctx);
  // (b) on the CU-masked update stream: the tile Cholesky of the look-ahead needs a whole CU (potrf.hip)
  hipStream_t sP = ctx->s_main, sU = ctx->s_upd;
  LPGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), sP));
  const int LTr = cyc_before(G.R, T), LTc = cyc_before(G.C, T);
  const bool la = ctx->lookahead != 0;

  // panels in order: every old panel (block append: solve + update of the new rows only), then the new ones
  struct Panel { int c0, c1; bool fresh; };
  std::vector<Panel> panels;
  if (t_done > 0 && T > t_done)
    for (int c0 = 0; c0 < t_done;) {
      const int c1 = std::min(t_done, (c0 / G.nbt + 1) * G.nbt);
      panels.push_back({c0, c1, false});
      c0 = c1;
    }
  for (int c0 = t_done; c0 < T;) {
    const int c1 = std::min(T, (c0 / G.nbt + 1) * G.nbt);
    panels.push_back({c0, c1, true});
    c0 = c1;
  }
  // rows a panel step touches: everything below the panel for a fresh panel, the new rows only for an old one
  auto row_lo = [&](const Panel& p) { return p.fresh ? p.c1 : t_done; };

  // Panel step, part 1 (panel stream): diagonal block, rows below, gather into panel buffer `which`
  // test hook: rank LPGP_TEST_FAIL_RANK fails (locally, like an allocation or launch error would) at its
  // LPGP_TEST_FAIL_PANEL-th panel step; tests/test_gpu_dist.py checks that the OTHER ranks return an error instead of
  // waiting for ever for that rank's pieces
  static const int fail_rank = [] { const char* e = std::getenv("LPGP_TEST_FAIL_RANK"); return e ? std::atoi(e) : -1; }();
  static const int fail_panel = [] { const char* e = std::getenv("LPGP_TEST_FAIL_PANEL"); return e ? std::atoi(e) : -1; }();
  int panel_steps = 0;
  // SPLIT GATHER (P x 1 grids, look-ahead on): the look-ahead update (a) of the next panel's columns reads, besides this
  // rank's OWN solved rows, only the rows of the next panel's diagonal block -- one small exchange from their owner.  So
  // the panel stream gathers just those (head), and the bulk of the panel (tail: S / P bytes per link, the longest item
  // of the chain while the panel is long, scratch/dist_model.py) travels on the exchange stream through the second
  // communicator, beside the chain, awaited only by the remainder update (b).
  const bool split = la && G.Pc == 1 && ctx->split_gather != 0 && ctx->world > 1;
  hipStream_t sC = sP;
  if (split && ctx->single_stream) {
    for (int i = 0; i < 2; ++i) {
      if (!ctx->ev_tail[i]) LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_tail[i], hipEventDisableTiming));
      if (!ctx->ev_rows[i]) LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_rows[i], hipEventDisableTiming));
    }
  } else if (split) {
    if (!ctx->s_comm) {
      int lo = 0, hi = 0;
      LPGP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      LPGP_HIP(hipStreamCreateWithPriority(&ctx->s_comm, hipStreamNonBlocking, hi));
      for (int i = 0; i < 2; ++i) {
        LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_tail[i], hipEventDisableTiming));
        LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_rows[i], hipEventDisableTiming));
      }
    }
    sC = ctx->s_comm;
  }
  bool tail_pending[2] = {false, false};
  // Panel step, part 1 (panel stream): diagonal block, rows below, gather into panel buffer `which`; head_end: first
  // global tile row the look-ahead update does NOT need (end of the next panel), T without a next panel
  auto panel_part = [&](const Panel& p, int which, int head_end) -> int {
    LPGP_CHECK(!(ctx->rank == fail_rank && panel_steps++ == fail_panel), "injected failure at panel step %d (LPGP_TEST_FAIL_PANEL)",
               fail_panel);
    const int pcK = (p.c0 / G.nbt) % G.Pc;
    if (p.fresh) LPGP_TRY(factor_diag_block(ctx, sP, mat, G, T, p.c0, p.c1));
    tail_pending[which] = false;
    if (p.c1 >= T) return 0;
    if (G.my_c == pcK) LPGP_TRY(panel_rows_solve(ctx, sP, mat, G, p.c0, p.c1, cyc_before(G.R, row_lo(p)), LTr));
    LPGP_TRY(ensure_panel(ctx, which, (size_t)(T - p.c1) * TILE * (size_t)(p.c1 - p.c0) * TILE));
    double* out = ctx->d_panel[which];
    if (!split) return gather_panel(ctx, sP, mat, G, T, p.c0, p.c1, p.c1, out, ctx->scoped_gather != 0);   // the updates read rows of this rank's row / column class only
    LPGP_HIP(hipEventRecord(ctx->ev_rows[which], sP));           // the rows below the panel are final
    LPGP_TRY(self_unpack(ctx, sP, mat, G, T, p.c0, p.c1, p.c1, out));
    LPGP_TRY(gather_rows(ctx, sP, mat, G, T, p.c0, p.c1, p.c1, p.c1, head_end, false, false, out));
    if (head_end < T) {
      LPGP_HIP(hipStreamWaitEvent(sC, ctx->ev_rows[which], 0));   // (also orders the tail behind the last readers of this buffer)
      LPGP_TRY(gather_rows(ctx, sC, mat, G, T, p.c0, p.c1, p.c1, head_end, T, false, true, out));
      LPGP_HIP(hipEventRecord(ctx->ev_tail[which], sC));
      tail_pending[which] = true;
    }
    return 0;
  };
  auto head_end_of = [&](size_t i) { return i + 1 < panels.size() ? panels[i + 1].c1 : T; };

  bool have_upd = false;
  hipEvent_t last_upd = nullptr;
  for (size_t i = 0; i < panels.size(); ++i) {
    const Panel& p = panels[i];
    const int which = (int)(i & 1);
    if (i == 0) {
      // the panel buffer about to be written was last read by the update two panels ago -- none yet
      LPGP_TRY(panel_part(p, which, head_end_of(i)));
    }
    if (p.c1 >= T) break;
    const double* panel = ctx->d_panel[which];
    const int64_t ldp = (int64_t)(T - p.c1) * TILE;
    const int K128 = (p.c1 - p.c0) * TILE;
    const int rt0 = cyc_before(G.R, row_lo(p));
    const int ct0 = cyc_before(G.C, p.c1);
    if (!la || i + 1 >= panels.size()) {
      if (tail_pending[which]) LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_tail[which], 0));
      LPGP_TRY(update_local(ctx, sP, mat, G, panel, ldp, p.c1, K128, rt0, LTr, ct0, LTc, LPGP_K_SYRK));
      if (i + 1 < panels.size()) LPGP_TRY(panel_part(panels[i + 1], which ^ 1, head_end_of(i + 1)));
      continue;
    }
    // look-ahead: (a) the next panel's own columns first, on the panel stream, followed at once by that panel's
    // factorisation / solve / gather; (b) everything to the right of it meanwhile on the update stream
    const Panel& q = panels[i + 1];
    const int cta = cyc_before(G.C, q.c0), ctb = cyc_before(G.C, q.c1);      // local columns of the next panel (empty off its process column)
    // The single-GPU scheduler's two rules (potrf.hip: factor_columns), priced with THIS rank's share of the update -- with P
    // ranks the remainder update is P times shorter and the chain (diagonal block, broadcast, rows, gather) longer, so the
    // chain bounds the pipeline from much earlier on: (1) while it does, (b) is released only when the look-ahead half (a) is
    // complete (launched together they share the chip by workgroup count and (a), which the next panel waits for, crawls);
    // (2) once (b) is shorter than the chain even on the narrow stream, it runs there and leaves a quarter of the CUs to the
    // chain's kernels.
    const double t_b_us = (double)gemm_valid_tiles(update_args(mat, G, panel, ldp, p.c1, K128, rt0, LTr, ctb, LTc, 1)) *
                          (2.0 * TILE * TILE * (double)K128 / 50e6);
    const double t_chain_us = ctx->chain_us_tile * (double)(q.c1 - q.c0) + ctx->chain_us_fixed + (ctx->world > 1 ? ctx->dist_chain_us_comm : 0.0);
    const bool chain_bound = t_b_us < t_chain_us;
    hipEvent_t evp = ctx->ev_panel[i & 1];
    if (!chain_bound) LPGP_HIP(hipEventRecord(evp, sP));                     // panel p is gathered (split: its head)
    if (have_upd) LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_upd[(i + 1) & 1], 0));   // (a)'s columns were last written by the previous (b)
    LPGP_TRY(update_local(ctx, sP, mat, G, panel, ldp, p.c1, K128, rt0, LTr, cta, ctb, LPGP_K_SYRK_AHEAD));
    if (chain_bound) LPGP_HIP(hipEventRecord(evp, sP));
    const double narrow_frac = ctx->cus > 0 ? (double)ctx->cus / (double)(ctx->cus - ctx->reserve_narrow) : 1.0;
    hipStream_t sB = (ctx->s_upd_narrow && t_b_us * narrow_frac < t_chain_us) ? ctx->s_upd_narrow : sU;
    if (last_upd) LPGP_HIP(hipStreamWaitEvent(sB, last_upd, 0));             // behind the previous remainder update (it may have run on the other update stream)
    LPGP_HIP(hipStreamWaitEvent(sB, evp, 0));
    if (tail_pending[which]) LPGP_HIP(hipStreamWaitEvent(sB, ctx->ev_tail[which], 0));   // (b) reads every row of the panel
    {
      GemmArgs gb = update_args(mat, G, panel, ldp, p.c1, K128, rt0, LTr, ctb, LTc, 1);
      gb.occ3 = t_b_us > ctx->gemm3_margin * t_chain_us;
      if (gb.mt > 0 && gb.nt > 0) LPGP_TRY(launch_gemm(ctx, sB, 0, 0, gb, LPGP_K_SYRK));
    }
    LPGP_HIP(hipEventRecord(ctx->ev_upd[i & 1], sB));
    last_upd = ctx->ev_upd[i & 1];
    have_upd = true;
    // the next gather writes panel buffer which ^ 1, last read by (b) of panel i - 1: that update has been waited
    // for above (ev_upd[(i + 1) & 1]) before anything of this step was enqueued on the panel stream
    LPGP_TRY(panel_part(q, which ^ 1, head_end_of(i + 1)));
  }
  if (have_upd && last_upd) LPGP_HIP(hipStreamWaitEvent(sP, last_upd, 0));      // join: every update is behind the panel stream
  LPGP_HIP(hipMemcpyAsync(ctx->h_info_pinned + 1, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, sP));
  LPGP_TRY(sync_stream(ctx, sP));
  int h_info = ctx->h_info_pinned[1];
  LPGP_TRY(allreduce_max_int(ctx, sP, &h_info));        // a failed pivot is seen by the owner of its diagonal block only
  if (info) *info = h_info;
  return 0;
}

// ---- solves: the factor is streamed, panel by panel, against this rank's columns of the right-hand side ---------
// V (all T * 128 rows x m_pad columns of THIS rank, column-major, leading dimension ldv) <- L^{-1} V
int trsm_lower_dist(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T64, double* v, int64_t ldv, int64_t m_pad) {
  const int T = (int)T64;
  const Grid G = grid_of(ctx);
  const int mtl = (int)(m_pad / TILE);
  const int64_t nb = (int64_t)G.nbt * TILE;
  const bool la = ctx->lookahead != 0 && mtl >= 8;       // worth it only for wide right-hand sides
  hipStream_t sP = ctx->s_main, sU = la ? ctx->s_upd_all : ctx->s_main;
  struct Panel { int c0, c1; };
  std::vector<Panel> panels;
  for (int c0 = 0; c0 < T;) {
    const int c1 = std::min(T, (c0 / G.nbt + 1) * G.nbt);
    panels.push_back({c0, c1});
    c0 = c1;
  }
  // panel stream: gather the rows below panel p into buffer `which` (it does not depend on the tile steps and
  // travels underneath them), then the tile steps of the panel's own rows
  auto panel_part = [&](const Panel& p, int which) -> int {
    const int K = p.c0 / G.nbt, b0 = p.c0 - K * G.nbt, kw = p.c1 - p.c0;
    if (p.c1 < T) {
      LPGP_TRY(ensure_panel(ctx, which, (size_t)(T - p.c1) * TILE * (size_t)kw * TILE));
      LPGP_TRY(gather_panel(ctx, sP, mat, G, T, p.c0, p.c1, p.c1, ctx->d_panel[which]));
    }
    if (ctx->fused_solve && kw <= 4)
      return launch_trsv_panel(ctx, sP, v + (int64_t)p.c0 * TILE, ldv, mat->linv + (int64_t)p.c0 * TILE * TILE, dblk_tile(mat, G, K, b0, b0), nb,
                               kw, mtl, LPGP_K_PANEL);
    for (int j = 0; j < kw; ++j) {
      double* Vj = v + (int64_t)(p.c0 + j) * TILE;
      LPGP_TRY(launch_trsv_tile(ctx, sP, Vj, ldv, mat->linv + (int64_t)(p.c0 + j) * TILE * TILE, dblk_tile(mat, G, K, b0 + j, b0 + j), nb, mtl,
                                LPGP_K_TRSM));
      if (j + 1 < kw)
        LPGP_TRY(launch_gemm(ctx, sP, 0, 1,
                             mk(dblk_tile(mat, G, K, b0 + j + 1, b0 + j), nb, Vj, ldv, Vj + TILE, ldv, kw - j - 1, mtl, TILE, -1.0, 1.0, 0),
                             LPGP_K_GEMM));
    }
    return 0;
  };
  // rows [r0, r1) (global tiles, all below panel p) of V -= L[rows, panel p] V_p, from panel buffer `which`
  auto update_rows = [&](hipStream_t st, const Panel& p, int which, int r0, int r1) -> int {
    if (r1 <= r0) return 0;
    const int64_t ldp = (int64_t)(T - p.c1) * TILE;
    return launch_gemm(ctx, st, 0, 1,
                       mk(ctx->d_panel[which] + (int64_t)(r0 - p.c1) * TILE, ldp, v + (int64_t)p.c0 * TILE, ldv, v + (int64_t)r0 * TILE, ldv,
                          r1 - r0, mtl, (p.c1 - p.c0) * TILE, -1.0, 1.0, 0),
                       LPGP_K_GEMM);
  };
  LPGP_TRY(panel_part(panels[0], 0));
  bool have_upd = false;
  for (size_t i = 0; i + 1 < panels.size(); ++i) {
    const Panel& p = panels[i];
    const Panel& q = panels[i + 1];
    const int which = (int)(i & 1);
    if (!la) {
      LPGP_TRY(update_rows(sP, p, which, p.c1, T));
      LPGP_TRY(panel_part(q, which ^ 1));
      continue;
    }
    // look-ahead as in potrf_dist: (a) the next panel's rows on the panel stream, followed by that panel's gather and
    // tile steps; (b) the rows below meanwhile on the update stream
    hipEvent_t evp = ctx->ev_panel[i & 1];
    LPGP_HIP(hipEventRecord(evp, sP));
    if (have_upd) LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_upd[(i + 1) & 1], 0));   // (a)'s rows and the next panel buffer: last touched by the previous (b)
    LPGP_TRY(update_rows(sP, p, which, q.c0, q.c1));
    LPGP_HIP(hipStreamWaitEvent(sU, evp, 0));
    LPGP_TRY(update_rows(sU, p, which, q.c1, T));
    LPGP_HIP(hipEventRecord(ctx->ev_upd[i & 1], sU));
    have_upd = true;
    LPGP_TRY(panel_part(q, which ^ 1));
  }
  if (have_upd) {
    LPGP_HIP(hipEventRecord(ctx->ev_upd[0], sU));
    LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_upd[0], 0));
  }
  return 0;
}

// V <- L^{-T} V, same layout: panels in reverse order; x_K = L_KK^{-T} (y_K - L[below, K]^T x_below)
int trsm_lower_t_dist(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T64, double* v, int64_t ldv, int64_t m_pad) {
  const int T = (int)T64;
  const Grid G = grid_of(ctx);
  const int mtl = (int)(m_pad / TILE);
  const int64_t nb = (int64_t)G.nbt * TILE, tb = TILE;
  hipStream_t st = ctx->s_main;
  void* sp = nullptr;
  const size_t sbytes = (size_t)TILE * (size_t)m_pad * sizeof(double);
  if (pool_alloc(ctx, &sp, sbytes, nullptr) != 0) return -1;
  double* S = (double*)sp;
  struct Release { lpgp_ctx* c; void* p; size_t b; ~Release() { pool_free(c, p, b); } } release{ctx, sp, sbytes};
  std::vector<int> starts;
  for (int c0 = 0; c0 < T; c0 = std::min(T, (c0 / G.nbt + 1) * G.nbt)) starts.push_back(c0);
  for (int pi = (int)starts.size() - 1; pi >= 0; --pi) {
    const int c0 = starts[pi], c1 = std::min(T, (c0 / G.nbt + 1) * G.nbt);
    const int K = c0 / G.nbt, b0 = c0 - K * G.nbt;
    if (c1 < T) {
      LPGP_TRY(ensure_panel(ctx, 0, (size_t)(T - c1) * TILE * (size_t)(c1 - c0) * TILE));
      LPGP_TRY(gather_panel(ctx, st, mat, G, T, c0, c1, c1, ctx->d_panel[0]));
      // y_K -= L[below, K]^T x_below   (contraction over the rows below)
      LPGP_TRY(launch_gemm(ctx, st, 1, 1,
                           mk(ctx->d_panel[0], (int64_t)(T - c1) * TILE, v + (int64_t)c1 * TILE, ldv, v + (int64_t)c0 * TILE, ldv, c1 - c0, mtl,
                              (T - c1) * TILE, -1.0, 1.0, 0),
                           LPGP_K_GEMM));
    }
    for (int j = c1 - c0 - 1; j >= 0; --j) {
      double* Vj = v + (int64_t)(c0 + j) * TILE;
      const double* linv = mat->linv + (int64_t)(c0 + j) * TILE * TILE;
      const double* Ljj = dblk_tile(mat, G, K, b0 + j, b0 + j);
      // refined tile step (see trsm_lower_t_blocked): S = Linv^T y;  y <- y - L^T S;  S <- S + Linv^T y;  y <- S
      LPGP_TRY(launch_gemm(ctx, st, 1, 1, mk(linv, tb, Vj, ldv, S, tb, 1, mtl, TILE, 1.0, 0.0, 0), LPGP_K_TRSM));
      LPGP_TRY(launch_gemm(ctx, st, 1, 1, mk(Ljj, nb, S, tb, Vj, ldv, 1, mtl, TILE, -1.0, 1.0, 0), LPGP_K_TRSM));
      LPGP_TRY(launch_gemm(ctx, st, 1, 1, mk(linv, tb, Vj, ldv, S, tb, 1, mtl, TILE, 1.0, 1.0, 0), LPGP_K_TRSM));
      LPGP_TRY(copy2d(st, Vj, ldv, S, tb, tb, m_pad));
      if (j > 0)      // rows of the block above: y_i -= L[j, i]^T x_j  for the tiles i < j of the block
        LPGP_TRY(launch_gemm(ctx, st, 1, 1,
                             mk(dblk_tile(mat, G, K, b0 + j, b0), nb, Vj, ldv, v + (int64_t)c0 * TILE, ldv, j, mtl, TILE, -1.0, 1.0, 0),
                             LPGP_K_GEMM));
    }
  }
  return 0;
}

// the whole factor, padded, column-major pn x pn on the host of EVERY rank (tests, `gram.cholesky()`)
int factor_to_host_dist(lpgp_ctx* ctx, lpgp_mat* mat, double* out) {
  const int T = (int)(mat->pn / TILE);
  const int64_t pn = mat->pn;
  const Grid G = grid_of(ctx);
  hipStream_t st = ctx->s_main;
  std::fill(out, out + (size_t)pn * pn, 0.0);
  for (int c0 = 0; c0 < T;) {
    const int c1 = std::min(T, (c0 / G.nbt + 1) * G.nbt);
    // rows from the panel's own first tile on (the diagonal block included: gathered like any other rows)
    LPGP_TRY(ensure_panel(ctx, 0, (size_t)(T - c0) * TILE * (size_t)(c1 - c0) * TILE));
    LPGP_TRY(gather_panel(ctx, st, mat, G, T, c0, c1, c0, ctx->d_panel[0]));
    LPGP_TRY(sync_stream(ctx, st));
    const int64_t rows = (int64_t)(T - c0) * TILE;
    LPGP_HIP(hipMemcpy2D(out + (int64_t)c0 * TILE * (pn + 1), (size_t)pn * sizeof(double), ctx->d_panel[0], (size_t)rows * sizeof(double),
                         (size_t)rows * sizeof(double), (size_t)(c1 - c0) * TILE, hipMemcpyDeviceToHost));
    c0 = c1;
  }
  return 0;
}

}  // namespace lpgp
