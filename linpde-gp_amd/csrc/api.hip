// C-ABI entry points of liblpgp.so (declared in include/lpgp.h).

#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include <rccl/rccl.h>

#include "lpgp_internal.h"

namespace lpgp {

// ---------------------------------------------------------------------------------------
// profiling with HIP events on the launching stream
// ---------------------------------------------------------------------------------------
static hipEvent_t get_event(lpgp_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void prof_begin(lpgp_ctx* ctx, hipStream_t stream, int kernel, double flops, double bytes) {
  ctx->prof_open = 0;
  if (!(ctx->prof_on & (1 << kernel))) return;
  ctx->prof_open = 1;
  PendingEvent p;
  p.e0 = get_event(ctx);
  p.e1 = get_event(ctx);
  p.kernel = kernel;
  (void)hipEventRecord(p.e0, stream);
  ctx->pending.push_back(p);
  ctx->prof[kernel].launches += 1;
  ctx->prof[kernel].flops += flops;
  ctx->prof[kernel].bytes += bytes;
}

void prof_end(lpgp_ctx* ctx, hipStream_t stream) {
  if (!ctx->prof_open) return;
  ctx->prof_open = 0;
  (void)hipEventRecord(ctx->pending.back().e1, stream);
}

int prof_collect(lpgp_ctx* ctx) {
  if (ctx->pending.empty()) return 0;
  LPGP_HIP(hipDeviceSynchronize());
  for (auto& p : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) ctx->prof[p.kernel].ms += ms;
    ctx->event_pool.push_back(p.e0);
    ctx->event_pool.push_back(p.e1);
  }
  ctx->pending.clear();
  return 0;
}

// ---------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------
__global__ void set_identity_kernel(double* a, int64_t ld, int64_t from, int64_t to, Layout2D lay) {
  int64_t i = from + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= to) return;
  const int64_t lr = cyc_local(lay.rows, i), lc = cyc_local(lay.cols, i);
  if (lr >= 0 && lc >= 0) a[lr + lc * ld] = 1.0;
}

// out[j] = sum_i K[i + j*ld] * (w ? w[i] : K[i + j*ld])   over i < rows; one block per column
__global__ __launch_bounds__(256) void col_reduce_kernel(const double* __restrict__ K, int64_t ld, int64_t rows,
                                                          const double* __restrict__ w, double* __restrict__ out) {
  __shared__ double red[4];
  const double* col = K + (int64_t)blockIdx.x * ld;
  double acc = 0.0;
  for (int64_t i = threadIdx.x * 2; i < rows; i += 512) {
    double2 v = *reinterpret_cast<const double2*>(col + i);
    if (w) {
      double2 ww = *reinterpret_cast<const double2*>(w + i);
      acc = fma(v.x, ww.x, acc);
      acc = fma(v.y, ww.y, acc);
    } else {
      acc = fma(v.x, v.x, acc);
      acc = fma(v.y, v.y, acc);
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// both read-outs of the solved cross-covariance V = L^{-1} K in one pass:
//   out_dot[j] = sum_i V[i,j] z[i]   (posterior mean term  K_xX G^{-1} r = V^T z,  z = L^{-1} r)
//   out_sq[j]  = sum_i V[i,j]^2      (variance reduction)
__global__ __launch_bounds__(256) void col_reduce2_kernel(const double* __restrict__ V, int64_t ld, int64_t rows,
                                                           const double* __restrict__ z, double* __restrict__ out_dot,
                                                           double* __restrict__ out_sq) {
  __shared__ double red[8];
  const double* col = V + (int64_t)blockIdx.x * ld;
  double ad = 0.0, as = 0.0;
  for (int64_t i = threadIdx.x * 2; i < rows; i += 512) {
    const double2 v = *reinterpret_cast<const double2*>(col + i);
    const double2 zz = *reinterpret_cast<const double2*>(z + i);
    ad = fma(v.x, zz.x, ad);
    ad = fma(v.y, zz.y, ad);
    as = fma(v.x, v.x, as);
    as = fma(v.y, v.y, as);
  }
  for (int off = 32; off > 0; off >>= 1) {
    ad += __shfl_down(ad, off);
    as += __shfl_down(as, off);
  }
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = ad;
    red[4 + (threadIdx.x >> 6)] = as;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out_dot[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    out_sq[blockIdx.x] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}

int pool_alloc(lpgp_ctx* ctx, void** out, size_t bytes, bool* fresh) {
  int best = -1;
  for (int i = 0; i < (int)ctx->pool.size(); ++i) {
    const auto& b = ctx->pool[i];
    // (slack of half the request + 64 KB: with 1 MB, an 18-KB request took the 1.2-MB buffer of the tile inverses of a small matrix,
    //  whose own request then went to hipMalloc, and the overflowing pool to hipFree -- 0.18 ms per step at N_tot = 1 152)
    if (b.bytes >= bytes && b.bytes <= bytes + bytes / 2 + (64 << 10) && (best < 0 || b.bytes < ctx->pool[best].bytes)) best = i;
  }
  if (best >= 0) {
    *out = ctx->pool[best].p;
    ctx->pool.erase(ctx->pool.begin() + best);
    if (fresh) *fresh = false;
    return 0;
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) {
    // release cached buffers and retry once
    for (auto& b : ctx->pool) (void)hipFree(b.p);
    ctx->pool.clear();
    e = hipMalloc(out, bytes);
  }
  if (e != hipSuccess) {
    set_error("device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return -1;
  }
  if (fresh) *fresh = true;
  return 0;
}

void pool_free(lpgp_ctx* ctx, void* p, size_t bytes) {
  if (!p) return;
  if (ctx->pool.size() >= 12) {
    // drop the smallest cached buffer
    int sm = 0;
    for (int i = 1; i < (int)ctx->pool.size(); ++i)
      if (ctx->pool[i].bytes < ctx->pool[sm].bytes) sm = i;
    (void)hipFree(ctx->pool[sm].p);
    ctx->pool.erase(ctx->pool.begin() + sm);
  }
  ctx->pool.push_back({p, bytes});
}

__global__ void clear_rows_kernel(double* a, int64_t ld, int64_t r0, int64_t nr, int64_t c0, int64_t nc, Layout2D lay) {
  // zero the strip rows [r0, r0+nr) x cols [c0, c0+nc) of the GLOBAL padded matrix (the part this rank owns)
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t c = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (i >= nr || c >= nc) return;
  const int64_t lr = cyc_local(lay.rows, r0 + i), lc = cyc_local(lay.cols, c0 + c);
  if (lr >= 0 && lc >= 0) a[lr + lc * ld] = 0.0;
}
// identity tail of a block in ONE launch (round 5; three launches until then: a chain of small conditionings is launch-bound):
// pad rows [r0, r0 + pad) x columns [0, r0) <- 0;  rows [r0, cap) x pad columns [r0, r0 + pad) <- 0, ones on the diagonal
__global__ void pad_block_kernel(double* a, int64_t ld, int64_t r0, int64_t pad, int64_t cap, Layout2D lay) {
  const int64_t c = blockIdx.y + (int64_t)blockIdx.z * 65535;          // columns [0, r0 + pad)
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= r0 + pad) return;
  const int64_t row = r0 + i;
  if (row >= (c < r0 ? r0 + pad : cap)) return;
  const int64_t lr = cyc_local(lay.rows, row), lc = cyc_local(lay.cols, c);
  if (lr >= 0 && lc >= 0) a[lr + lc * ld] = (row == c) ? 1.0 : 0.0;
}
static void launch_pad_block(hipStream_t st, double* a, int64_t ld, int64_t r0, int64_t pad, int64_t cap, const Layout2D& lay) {
  const int64_t nc = r0 + pad, nr = cap - r0;
  const unsigned gy = (unsigned)(nc < 65535 ? nc : 65535), gz = (unsigned)((nc + 65534) / 65535);
  hipLaunchKernelGGL(pad_block_kernel, dim3((unsigned)((nr + 255) / 256), gy, gz), dim3(256), 0, st, a, ld, r0, pad, cap, lay);
}
// ... and, for lpgp_mat_condition, together with the block's measurement noise on the diagonal (sigma^2 I or a vector): ONE launch
// behind the assembly instead of a padding launch in front of it and an add_diag launch behind it.  The padding is the
// arithmetic of pad_block_kernel, the noise that of add_diag_kernel (one addition per diagonal entry: identical bits).
__global__ void finish_block_kernel(double* a, int64_t ld, int64_t r0, int64_t pad, int64_t cap, Layout2D lay, int64_t ncols, int64_t d0, int64_t dn,
                                    const double* dv, double dscalar) {
  const int64_t c = blockIdx.y + (int64_t)blockIdx.z * 65535;          // columns [0, ncols): padding; slice ncols: the noise
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c == ncols) {
    if (i < dn) {
      const int64_t lr = cyc_local(lay.rows, d0 + i), lc = cyc_local(lay.cols, d0 + i);
      if (lr >= 0 && lc >= 0) a[lr + lc * ld] += (dv ? dv[i] : 0.0) + dscalar;
    }
    return;
  }
  if (c > ncols) return;
  const int64_t row = r0 + i;
  if (row >= (c < r0 ? r0 + pad : cap)) return;
  const int64_t lr = cyc_local(lay.rows, row), lc = cyc_local(lay.cols, c);
  if (lr >= 0 && lc >= 0) a[lr + lc * ld] = (row == c) ? 1.0 : 0.0;
}
// pad == 0: noise only; dn == 0: padding only
static void launch_finish_block(hipStream_t st, double* a, int64_t ld, int64_t r0, int64_t pad, int64_t cap, const Layout2D& lay, int64_t d0, int64_t dn,
                                const double* dv, double dscalar) {
  const int64_t ncols = pad > 0 ? r0 + pad : 0, nr = pad > 0 ? cap - r0 : 0;
  if (ncols == 0 && dn == 0) return;
  const int64_t rows = nr > dn ? nr : dn, slices = ncols + (dn > 0 ? 1 : 0);
  const unsigned gy = (unsigned)(slices < 65535 ? slices : 65535), gz = (unsigned)((slices + 65534) / 65535);
  hipLaunchKernelGGL(finish_block_kernel, dim3((unsigned)((rows + 255) / 256), gy, gz), dim3(256), 0, st, a, ld, r0, pad, cap, lay, ncols, d0, dn, dv,
                     dscalar);
}
static void launch_clear_rows(hipStream_t st, double* a, int64_t ld, int64_t r0, int64_t nr, int64_t c0, int64_t nc, const Layout2D& lay) {
  if (nr <= 0 || nc <= 0) return;
  const unsigned gy = (unsigned)(nc < 65535 ? nc : 65535), gz = (unsigned)((nc + 65534) / 65535);
  hipLaunchKernelGGL(clear_rows_kernel, dim3((unsigned)((nr + 255) / 256), gy, gz), dim3(256), 0, st, a, ld, r0, nr, c0, nc, lay);
}

static int ensure_tmp(lpgp_ctx* ctx, int64_t n) {
  if (n <= ctx->tmp_cap) return 0;
  if (ctx->d_tmp) LPGP_HIP(hipFree(ctx->d_tmp));
  ctx->d_tmp = nullptr;
  ctx->tmp_cap = 0;
  LPGP_HIP(hipMalloc(&ctx->d_tmp, (size_t)n * sizeof(double)));
  ctx->tmp_cap = n;
  return 0;
}

static int ensure_stage(lpgp_ctx* ctx, int64_t n) {
  if (n <= ctx->stage_cap) return 0;
  if (ctx->h_stage) LPGP_HIP(hipHostFree(ctx->h_stage));
  ctx->h_stage = nullptr;
  ctx->stage_cap = 0;
  LPGP_HIP(hipHostMalloc(&ctx->h_stage, (size_t)n * sizeof(double), hipHostMallocDefault));
  ctx->stage_cap = n;
  return 0;
}

// layout of the matrix of this context: identity on a single GPU, the rank's share of the Pr x Pc grid otherwise
static Layout2D mat_layout(const lpgp_ctx* ctx) { return ctx->distributed() ? ctx->layout() : Layout2D(); }
static int64_t local_rows(const lpgp_ctx* ctx, int64_t padded) { return (int64_t)cyc_before(mat_layout(ctx).rows, (int)(padded / TILE)) * TILE; }
static int64_t local_cols(const lpgp_ctx* ctx, int64_t padded) { return (int64_t)cyc_before(mat_layout(ctx).cols, (int)(padded / TILE)) * TILE; }

static size_t mat_bytes_a(const lpgp_mat* m) { return (size_t)m->lr_cap * m->lc_cap * sizeof(double); }
static size_t mat_bytes_l(int64_t cap) { return (size_t)cap * TILE * sizeof(double); }
static size_t mat_bytes_w(int64_t cap) { return (size_t)(2 * cap + 8) * sizeof(double); }   // w | r | status word of the enqueued factorisations (lpgp_mat::d_status)
static size_t mat_bytes_d(const lpgp_ctx* ctx, int64_t cap) { return (size_t)round_up(cap, ctx->nb) * ctx->nb * sizeof(double); }

static void mat_release(lpgp_ctx* ctx, lpgp_mat* mat) {
  pool_free(ctx, mat->a, mat_bytes_a(mat));
  pool_free(ctx, mat->linv, mat_bytes_l(mat->cap));
  pool_free(ctx, mat->w, mat_bytes_w(mat->cap));
  if (mat->dblk) pool_free(ctx, mat->dblk, mat_bytes_d(ctx, mat->cap));
  mat->a = mat->linv = mat->w = mat->dblk = nullptr;
}

static int mat_alloc(lpgp_ctx* ctx, lpgp_mat* mat, int64_t cap) {
  // (re)allocate to `cap` (multiple of TILE) keeping the content of the first mat->pn rows/cols (of this rank's
  // share: the local index of a tile does not depend on the capacity).
  // Invariant kept for every buffer handed out: nothing is assumed about the lower triangle
  // of logical blocks (assembly overwrites it) nor about tiles above the diagonal (never
  // read); the padding strips of a block are cleared in lpgp_mat_add_block.
  const Layout2D lay = mat_layout(ctx);
  int64_t lr = local_rows(ctx, cap), lc = local_cols(ctx, cap);
  if (lr < TILE) lr = TILE;
  if (lc < TILE) lc = TILE;
  lpgp_mat fresh_dims;
  fresh_dims.lr_cap = lr; fresh_dims.lc_cap = lc;
  void *na = nullptr, *nl = nullptr, *nw = nullptr, *nd = nullptr;
  int rc = pool_alloc(ctx, &na, mat_bytes_a(&fresh_dims), nullptr);
  if (rc == 0) rc = pool_alloc(ctx, &nl, mat_bytes_l(cap), nullptr);
  if (rc == 0) rc = pool_alloc(ctx, &nw, mat_bytes_w(cap), nullptr);
  if (rc == 0 && ctx->distributed()) rc = pool_alloc(ctx, &nd, mat_bytes_d(ctx, cap), nullptr);
  if (rc != 0) return rc;
  const int64_t pn_all = mat->hidden.empty() ? mat->pn : mat->hidden.back().poff + mat->hidden.back().pn;
  if (mat->a && pn_all > 0) {
    const int64_t lru = local_rows(ctx, pn_all), lcu = local_cols(ctx, pn_all);
    if (lru > 0 && lcu > 0)
      LPGP_HIP(hipMemcpy2DAsync(na, (size_t)lr * sizeof(double), mat->a, (size_t)mat->lr_cap * sizeof(double),
                                (size_t)lru * sizeof(double), (size_t)lcu, hipMemcpyDeviceToDevice, ctx->s_main));
    LPGP_HIP(hipMemcpyAsync(nl, mat->linv, (size_t)pn_all * TILE * sizeof(double), hipMemcpyDeviceToDevice,
                            ctx->s_main));
    LPGP_HIP(hipMemcpyAsync(nw, mat->w, (size_t)pn_all * sizeof(double), hipMemcpyDeviceToDevice, ctx->s_main));
    LPGP_HIP(hipMemcpyAsync((double*)nw + 2 * cap, mat->d_status, sizeof(double), hipMemcpyDeviceToDevice, ctx->s_main));   // (sticky status: kept)
    if (nd) LPGP_HIP(hipMemcpyAsync(nd, mat->dblk, mat_bytes_d(ctx, pn_all), hipMemcpyDeviceToDevice, ctx->s_main));
    // padding columns of the existing blocks must stay zero in the rows added by the growth
    for (const auto& b : mat->blocks) {
      const int64_t padc = b.pn - b.n;
      if (padc > 0) launch_clear_rows(ctx->s_main, (double*)na, lr, pn_all, cap - pn_all, b.poff + b.n, padc, lay);
    }
  }
  if (!(mat->a && pn_all > 0)) LPGP_HIP(hipMemsetAsync((double*)nw + 2 * cap, 0, sizeof(double), ctx->s_main));       // a new / empty matrix: status clear
  LPGP_HIP(hipStreamSynchronize(ctx->s_main));
  if (mat->a) mat_release(ctx, mat);
  mat->a = (double*)na;
  mat->linv = (double*)nl;
  mat->w = (double*)nw;
  mat->d_status = reinterpret_cast<int*>((double*)nw + 2 * cap);
  mat->dblk = (double*)nd;
  mat->cap = cap;
  mat->lr_cap = lr;
  mat->lc_cap = lc;
  mat->has_r = 0;            // the residual segment is addressed relative to cap
  return 0;
}

// logical (length mat->n) <-> padded (length mat->pn) host vectors
static void scatter_padded(const lpgp_mat* mat, const double* logical, double* padded) {
  std::memset(padded, 0, (size_t)mat->pn * sizeof(double));
  for (const auto& b : mat->blocks) std::memcpy(padded + b.poff, logical + b.off, (size_t)b.n * sizeof(double));
}
static void gather_padded(const lpgp_mat* mat, const double* padded, double* logical) {
  for (const auto& b : mat->blocks) std::memcpy(logical + b.off, padded + b.poff, (size_t)b.n * sizeof(double));
}

}  // namespace lpgp

using namespace lpgp;

extern "C" {

const char* lpgp_last_error(void) { return lpgp::last_error(); }

int lpgp_init(int device, lpgp_ctx** out) {
  LPGP_CHECK(out != nullptr, "lpgp_init: null out");
  int ndev = 0;
  LPGP_HIP(hipGetDeviceCount(&ndev));
  LPGP_CHECK(ndev > 0, "lpgp_init: no HIP device visible (this library has no CPU fallback)");
  LPGP_CHECK(device >= 0 && device < ndev, "lpgp_init: device %d out of range (%d visible)", device, ndev);
  LPGP_HIP(hipSetDevice(device));
  lpgp_ctx* ctx = new lpgp_ctx();
  ctx->device = device;
  hipDeviceProp_t prop;
  LPGP_HIP(hipGetDeviceProperties(&prop, device));
  ctx->cus = prop.multiProcessorCount;
  int lo = 0, hi = 0;
  LPGP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  LPGP_HIP(hipStreamCreateWithPriority(&ctx->s_main, hipStreamNonBlocking, hi));
  // The trailing-update stream may use all CUs except a few reserved ones, so that the
  // panel kernels of the look-ahead (a 156-KB-LDS tile Cholesky needs a whole CU) never queue
  // behind thousands of resident update workgroups.  LPGP_RESERVE_CUS=0 disables the mask.
  // 32 = four CUs per XCD.  In steady state the mask costs the rank-512 update 10 % whether 1, 8 or 32 CUs are missing (65.1 ->
  // 58.8 / 58.5 TFLOP/s: the dispatcher feeds the shader engines evenly, so the first missing CU of an engine already sets
  // its pace; 64 missing cost 22 %; profiles/r03_clock_power.txt) -- so the chain may as well have the 32: c3 55.03-55.21 ms
  // against 55.18-55.64 with 8, c2 / c4 / c5 within their spread (MEASUREMENTS.md).
  int reserve = 32;
  if (const char* e = std::getenv("LPGP_RESERVE_CUS")) reserve = std::atoi(e);
  if (const char* e = std::getenv("LPGP_RESERVE_CUS_NARROW")) ctx->reserve_narrow = std::atoi(e);
  // mask bit i = CU (i / 8) of XCD (i % 8) (measured, scratch/cumask.hip): clearing the low
  // `reserve` bits takes the CUs round-robin from the XCDs, reserve / 8 per XCD,
  // so a single workgroup of the panel stream finds a free CU on whichever XCD it is dealt to
  auto masked_stream = [&](int nreserve, hipStream_t* out) {
    *out = nullptr;
    if (nreserve == 0 || nreserve >= ctx->cus) return;       // (negative: a CU-masked queue with every CU enabled, for measurements)
    const int words = (ctx->cus + 31) / 32;
    std::vector<uint32_t> mask(words, 0u);
    for (int cu = 0; cu < ctx->cus; ++cu) mask[cu / 32] |= (1u << (cu % 32));
    for (int r = 0; r < nreserve; ++r) mask[r / 32] &= ~(1u << (r % 32));
    if (hipExtStreamCreateWithCUMask(out, (uint32_t)words, mask.data()) != hipSuccess) {
      (void)hipGetLastError();
      *out = nullptr;
    }
  };
  // LPGP_SINGLE_STREAM=1: every stream of the context is the panel stream (no concurrency between chain and update).  For
  // jobs whose ranks SHARE one GPU (tests: eight processes x six queues oversubscribe the device's hardware queues and a
  // small eight-rank case takes 300 s; with one queue per process it takes a fraction of that).  Never a production setting.
  const bool single_stream = [] { const char* e = std::getenv("LPGP_SINGLE_STREAM"); return e && std::atoi(e) != 0; }();
  if (single_stream) {
    ctx->s_upd = ctx->s_upd_narrow = ctx->s_upd_all = ctx->s_outer = ctx->s_main;
    ctx->single_stream = 1;
  } else {
  masked_stream(reserve, &ctx->s_upd);
  if (!ctx->s_upd) LPGP_HIP(hipStreamCreateWithPriority(&ctx->s_upd, hipStreamNonBlocking, lo));
  // While the panel chain bounds the pipeline (small trailing matrix) the update can spare a
  // quarter of the chip: with only 8 reserved CUs the chain's TRSM / in-panel update wait for
  // update workgroups to retire (measured at panel 20 of c3: 92 + 154 us instead of 21 + 15).
  masked_stream(ctx->reserve_narrow, &ctx->s_upd_narrow);
  // (a CU mask costs more than its CUs: a 16384^2 x 512 SYRK runs at 53.5 TFLOP/s unmasked, 49.4
  //  with 8 CUs removed (tile-count quantisation on 496 instead of 512 slots), 44.3 with 64
  //  removed; LPGP_TEST_GEMM_STREAM + scratch/gemm_sweep.py.  Only the factorisation needs it.)
  LPGP_HIP(hipStreamCreateWithPriority(&ctx->s_upd_all, hipStreamNonBlocking, lo));
  masked_stream(reserve, &ctx->s_outer);
  if (!ctx->s_outer) LPGP_HIP(hipStreamCreateWithPriority(&ctx->s_outer, hipStreamNonBlocking, lo));
  }
  if (const char* e = std::getenv("LPGP_NB_OUTER")) {
    long v = std::atol(e);
    if (v >= 0 && v % TILE == 0) ctx->nb_outer = v;
  }
  if (const char* e = std::getenv("LPGP_NB_OUTER_MIN_TILES")) ctx->nb_outer_min_tiles = std::atoi(e);
  for (int i = 0; i < 2; ++i) {
    LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_outer[i], hipEventDisableTiming));
    LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_outer_fact[i], hipEventDisableTiming));
    LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_outer_a1[i], hipEventDisableTiming));
  }
  for (int i = 0; i < 4; ++i) LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_ride[i], hipEventDisableTiming));
  LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_chain_pre, hipEventDisableTiming));
  LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_chain_rows, hipEventDisableTiming));
  LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_append[0], hipEventDisableTiming));
  LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_append[1], hipEventDisableTiming));
  if (const char* e = std::getenv("LPGP_RIDE_STREAM")) ctx->ride_stream = std::atoi(e);
  if (const char* e = std::getenv("LPGP_CHAIN_RESIDENT")) ctx->chain_resident_max_rows = std::atoi(e);
  if (const char* e = std::getenv("LPGP_TRSV_RESIDENT")) ctx->trsv_resident = std::atoi(e);
  if (const char* e = std::getenv("LPGP_CHAIN_RESIDENT2")) ctx->chain_resident2_max_rows = std::atoi(e);
  if (const char* e = std::getenv("LPGP_CHAIN_AHEAD")) ctx->chain_ahead = std::atoi(e);
  if (const char* e = std::getenv("LPGP_CHAIN_AHEAD_MIN_ROWS")) ctx->chain_ahead_min_rows = std::atoi(e);
  if (const char* e = std::getenv("LPGP_RIDE_OCC3")) ctx->ride_occ3 = std::atoi(e);
  if (const char* e = std::getenv("LPGP_RIDE_AUG")) ctx->ride_aug = std::atoi(e);
  if (const char* e = std::getenv("LPGP_RIDE_B_ON_RIDE")) ctx->ride_b_on_ride = std::atoi(e);
  if (const char* e = std::getenv("LPGP_APPEND_SPLIT")) ctx->append_split = std::atoi(e);
  if (const char* e = std::getenv("LPGP_APPEND_SPLIT_MIN_TILES")) ctx->append_split_min_tiles = std::atoi(e);
  if (const char* e = std::getenv("LPGP_RIDE_OLD_UNGATED")) ctx->ride_old_ungated = std::atoi(e);
  // A profiler that SERIALISES kernels (rocprofv3 --pmc / counter groups: ROCPROF_COUNTER_COLLECTION) breaks the one assumption of
  // the follower -- that its chain kernel is dispatched beside it: it would wait out its poll limit, ~1 s per panel, and the step
  // would fail with a negative status.  Off there unless asked for explicitly.
  if (const char* e = std::getenv("ROCPROF_COUNTER_COLLECTION")) { if (e[0] != '\0' && e[0] != '0' && e[0] != 'F' && e[0] != 'f') ctx->ride_vchain_max_wgs = 0; }
  if (const char* e = std::getenv("LPGP_RIDE_VCHAIN")) ctx->ride_vchain_max_wgs = std::atoi(e);
  if (const char* e = std::getenv("LPGP_RIDE_VCHAIN_PRE")) ctx->ride_vchain_pre = std::atoi(e);
  if (const char* e = std::getenv("LPGP_RIDE_GATE_PCT")) ctx->ride_gate_pct = std::atoi(e);
  if (const char* e = std::getenv("LPGP_RIDE_OUTER_ROWS")) ctx->ride_outer_rows = std::atol(e);
  if (const char* e = std::getenv("LPGP_RIDE_OUTER_MIN_TILES")) ctx->ride_outer_min_tiles = std::atoi(e);
  if (const char* e = std::getenv("LPGP_RIDE_MAX_TILES")) ctx->ride_max_tiles = std::atoi(e);
  if (const char* e = std::getenv("LPGP_RIDE_SAME_STREAM_MAX_TILES")) ctx->ride_same_stream_max_tiles = std::atoi(e);
  for (int i = 0; i < 2; ++i) {
    LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_panel[i], hipEventDisableTiming));
    LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_upd[i], hipEventDisableTiming));
  }
  for (auto& sl : ctx->desc_ring) {
    LPGP_HIP(hipHostMalloc(&sl.h, sizeof(DevDesc), hipHostMallocDefault));
    LPGP_HIP(hipMalloc(&sl.d, sizeof(DevDesc)));
    LPGP_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  }
  LPGP_HIP(hipMalloc(&ctx->d_info, sizeof(int)));
  LPGP_HIP(hipHostMalloc(&ctx->h_info_pinned, 64, hipHostMallocDefault));
  if (const char* e = std::getenv("LPGP_NB")) {
    long v = std::atol(e);
    if (v >= TILE && v % TILE == 0) ctx->nb = v;
  }
  if (const char* e = std::getenv("LPGP_NB_OUTER_SOLVE")) {
    long v = std::atol(e);
    if (v >= 0 && v % TILE == 0) ctx->nb_outer_solve = v;
  }
  if (const char* e = std::getenv("LPGP_NB_OUTER_SOLVE_MIN_TILES")) ctx->nb_outer_solve_min_tiles = std::atoi(e);
  if (const char* e = std::getenv("LPGP_NB_SOLVE")) {
    long v = std::atol(e);
    if (v >= 0 && v % TILE == 0) ctx->nb_solve = v;
  }
  if (const char* e = std::getenv("LPGP_LOOKAHEAD")) ctx->lookahead = std::atoi(e) != 0;
  if (const char* e = std::getenv("LPGP_CHAIN_US_TILE")) ctx->chain_us_tile = std::atof(e);
  if (const char* e = std::getenv("LPGP_SOLVE_CHAIN_US_TILE")) ctx->solve_chain_us_tile = std::atof(e);
  if (const char* e = std::getenv("LPGP_CHAIN_US_FIXED")) ctx->chain_us_fixed = std::atof(e);
  if (const char* e = std::getenv("LPGP_DENSE_TILES")) ctx->dense_tiles = std::atoi(e) != 0;
  if (const char* e = std::getenv("LPGP_FUSED_SOLVE")) ctx->fused_solve = std::atoi(e) != 0;
  if (const char* e = std::getenv("LPGP_GEMM3")) ctx->gemm3 = std::atoi(e);
  if (const char* e = std::getenv("LPGP_SMALL_RING2")) ctx->small_ring2 = std::atoi(e);
  if (const char* e = std::getenv("LPGP_SMALL_TILES_MAX")) ctx->small_tiles_max = std::atoi(e);
  if (const char* e = std::getenv("LPGP_FUSED_AHEAD")) ctx->fused_ahead = std::atoi(e);
  if (const char* e = std::getenv("LPGP_PANEL_EXCLUSIVE")) ctx->panel_exclusive = std::atoi(e);
  if (const char* e = std::getenv("LPGP_FUSED_AHEAD_MIN_US")) ctx->fused_ahead_min_us = std::atoi(e);
  if (const char* e = std::getenv("LPGP_GEMM3_MARGIN")) ctx->gemm3_margin = std::atof(e);
  if (const char* e = std::getenv("LPGP_GEMM3_FACT")) ctx->gemm3_fact = std::atoi(e);
  if (const char* e = std::getenv("LPGP_ASM_FACTORS")) ctx->asm_factors = std::atoi(e) != 0;
  if (const char* e = std::getenv("LPGP_ASM_FAST")) ctx->asm_fast = std::atoi(e) != 0;
  if (const char* e = std::getenv("LPGP_ASM_CT")) ctx->asm_ct = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("LPGP_ASM_BATCH")) ctx->asm_batch = std::atoi(e) != 0;
  if (const char* e = std::getenv("LPGP_KRON_WIDE")) ctx->kron_wide = std::atoi(e) != 0;
  if (const char* e = std::getenv("LPGP_DIST_COLLECTIVE")) ctx->dist_bcast = std::strcmp(e, "bcast") == 0;
  if (const char* e = std::getenv("LPGP_DIST_SPLIT_GATHER")) ctx->split_gather = std::atoi(e) != 0;
  if (const char* e = std::getenv("LPGP_DIST_SCOPED_GATHER")) ctx->scoped_gather = std::atoi(e) != 0;
  if (const char* e = std::getenv("LPGP_DIST_CHAIN_US_COMM")) ctx->dist_chain_us_comm = std::atof(e);
  if (const char* e = std::getenv("LPGP_GEMM_BAND")) { const int v = std::atoi(e); if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32) ctx->gemm_band = v; }
  if (const char* e = std::getenv("LPGP_NB_BIG")) {
    long v = std::atol(e);
    if (v >= 0 && v % TILE == 0) ctx->nb_big = v;
  }
  if (const char* e = std::getenv("LPGP_NB_BIG_MIN_TILES")) ctx->nb_big_min_tiles = std::atoi(e);
  *out = ctx;
  return 0;
}

int lpgp_finalize(lpgp_ctx* ctx) {
  if (!ctx) return 0;
  (void)hipSetDevice(ctx->device);
  if (ctx->nccl_comm && ctx->world > 1 && sync_stream(ctx, ctx->s_main) != 0) {
    // (sync_stream has aborted the communicators: a kernel of this rank was still waiting for a peer that is gone)
  }
  (void)hipDeviceSynchronize();
  for (auto& p : ctx->pending) {
    (void)hipEventDestroy(p.e0);
    (void)hipEventDestroy(p.e1);
  }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ctx->ev_ride[i]);
  (void)hipEventDestroy(ctx->ev_chain_pre);
  if (ctx->ev_chain_rows) (void)hipEventDestroy(ctx->ev_chain_rows);
  for (auto& e : ctx->ev_append) if (e) (void)hipEventDestroy(e);
  for (int i = 0; i < 2; ++i) {
    (void)hipEventDestroy(ctx->ev_panel[i]);
    (void)hipEventDestroy(ctx->ev_upd[i]);
    (void)hipEventDestroy(ctx->ev_outer[i]);
    (void)hipEventDestroy(ctx->ev_outer_fact[i]);
    (void)hipEventDestroy(ctx->ev_outer_a1[i]);
  }
  for (auto& sl : ctx->desc_ring) {
    (void)hipHostFree(sl.h);
    (void)hipFree(sl.d);
    (void)hipEventDestroy(sl.done);
  }
  (void)hipFree(ctx->d_info);
  if (ctx->d_chain_flags) (void)hipFree(ctx->d_chain_flags);
  (void)hipHostFree(ctx->h_info_pinned);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  for (auto& b : ctx->pts_pool) (void)hipFree(b.p);
  ctx->pts_pool.clear();
  for (auto& sl : ctx->pts_ring) {
    if (sl.h) (void)hipHostFree(sl.h);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  if (ctx->h_stage_r) (void)hipHostFree(ctx->h_stage_r);
  if (ctx->ev_stage_r) (void)hipEventDestroy(ctx->ev_stage_r);
  if (ctx->d_tmp) (void)hipFree(ctx->d_tmp);
  for (size_t r = 0; r < ctx->ipc_peer.size(); ++r)
    if ((int)r != ctx->rank && ctx->ipc_peer[r]) (void)hipIpcCloseMemHandle(ctx->ipc_peer[r]);
  if (ctx->ipc_window) (void)hipFree(ctx->ipc_window);
  for (int i = 0; i < 2; ++i)
    if (ctx->ev_ipc[i]) (void)hipEventDestroy(ctx->ev_ipc[i]);
  if (ctx->d_pack) (void)hipFree(ctx->d_pack);
  for (int i = 0; i < 2; ++i)
    if (ctx->d_panel[i]) (void)hipFree(ctx->d_panel[i]);
  if (ctx->nccl_comm_bulk) (void)ncclCommDestroy((ncclComm_t)ctx->nccl_comm_bulk);
  if (ctx->nccl_comm) (void)ncclCommDestroy((ncclComm_t)ctx->nccl_comm);
  if (ctx->d_pack_bulk) (void)hipFree(ctx->d_pack_bulk);
  if (ctx->s_comm) (void)hipStreamDestroy(ctx->s_comm);
  for (int i = 0; i < 2; ++i) {
    if (ctx->ev_tail[i]) (void)hipEventDestroy(ctx->ev_tail[i]);
    if (ctx->ev_rows[i]) (void)hipEventDestroy(ctx->ev_rows[i]);
  }
  for (auto& b : ctx->pool) (void)hipFree(b.p);
  ctx->pool.clear();
  if (!ctx->single_stream) {
    (void)hipStreamDestroy(ctx->s_upd);
    if (ctx->s_upd_narrow) (void)hipStreamDestroy(ctx->s_upd_narrow);
    (void)hipStreamDestroy(ctx->s_upd_all);
    if (ctx->s_outer) (void)hipStreamDestroy(ctx->s_outer);
  }
  (void)hipStreamDestroy(ctx->s_main);
  delete ctx;
  return 0;
}

int lpgp_device_info(lpgp_ctx* ctx, char* name, int len, int* cus, int64_t* hbm_bytes) {
  LPGP_DEVICE(ctx);
  hipDeviceProp_t prop;
  LPGP_HIP(hipGetDeviceProperties(&prop, ctx->device));
  if (name && len > 0) {
    std::snprintf(name, (size_t)len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cus) *cus = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return 0;
}

int lpgp_sync(lpgp_ctx* ctx) {
  LPGP_HIP(hipSetDevice(ctx->device));
  LPGP_HIP(hipDeviceSynchronize());
  return 0;
}

int lpgp_get_option(lpgp_ctx* ctx, const char* key, int64_t* value) {
  LPGP_CHECK(ctx && key && value, "lpgp_get_option: null argument");
  if (std::strcmp(key, "nb") == 0) *value = ctx->nb;
  else if (std::strcmp(key, "gemm3") == 0) *value = ctx->gemm3;
  else if (std::strcmp(key, "gemm3_fact") == 0) *value = ctx->gemm3_fact;
  else if (std::strcmp(key, "dist_bcast") == 0) *value = ctx->dist_bcast;
  else if (std::strcmp(key, "split_gather") == 0) *value = ctx->split_gather;
  else if (std::strcmp(key, "scoped_gather") == 0) *value = ctx->scoped_gather;
  else if (std::strcmp(key, "lookahead") == 0) *value = ctx->lookahead;
  else if (std::strcmp(key, "fused_solve") == 0) *value = ctx->fused_solve;
  else if (std::strcmp(key, "small_tiles_max") == 0) *value = ctx->small_tiles_max;
  else if (std::strcmp(key, "live_mats") == 0) *value = ctx->live_mats;
  else if (std::strcmp(key, "asm_ct") == 0) *value = ctx->asm_ct;
  else if (std::strcmp(key, "asm_batch") == 0) *value = ctx->asm_batch;
  else if (std::strcmp(key, "kron_wide") == 0) *value = ctx->kron_wide;
  else if (std::strcmp(key, "asm_fast") == 0) *value = ctx->asm_fast;
  else if (std::strcmp(key, "fused_ahead") == 0) *value = ctx->fused_ahead;
  else if (std::strcmp(key, "fused_ahead_min_us") == 0) *value = ctx->fused_ahead_min_us;
  else if (std::strcmp(key, "nb_outer_solve") == 0) *value = ctx->nb_outer_solve;
  else if (std::strcmp(key, "nb_outer_solve_min_tiles") == 0) *value = ctx->nb_outer_solve_min_tiles;
  else if (std::strcmp(key, "nb_solve") == 0) *value = ctx->nb_solve;
  else if (std::strcmp(key, "solve_chain_us_tile") == 0) *value = (int64_t)ctx->solve_chain_us_tile;
  else if (std::strcmp(key, "chain_us_fixed") == 0) *value = (int64_t)ctx->chain_us_fixed;
  else if (std::strcmp(key, "ride_stream") == 0) *value = ctx->ride_stream;
  else if (std::strcmp(key, "chain_resident_max_rows") == 0) *value = ctx->chain_resident_max_rows;
  else if (std::strcmp(key, "trsv_resident") == 0) *value = ctx->trsv_resident;
  else if (std::strcmp(key, "chain_resident2_max_rows") == 0) *value = ctx->chain_resident2_max_rows;
  else if (std::strcmp(key, "chain_ahead") == 0) *value = ctx->chain_ahead;
  else if (std::strcmp(key, "chain_ahead_min_rows") == 0) *value = ctx->chain_ahead_min_rows;
  else if (std::strcmp(key, "ride_vchain_max_wgs") == 0) *value = ctx->ride_vchain_max_wgs;
  else if (std::strcmp(key, "ride_occ3") == 0) *value = ctx->ride_occ3;
  else if (std::strcmp(key, "ride_aug") == 0) *value = ctx->ride_aug;
  else if (std::strcmp(key, "append_split") == 0) *value = ctx->append_split;
  else if (std::strcmp(key, "ride_gate_pct") == 0) *value = ctx->ride_gate_pct;
  else if (std::strcmp(key, "ride_outer_rows") == 0) *value = ctx->ride_outer_rows;
  else if (std::strcmp(key, "ride_outer_min_tiles") == 0) *value = ctx->ride_outer_min_tiles;
  else if (std::strcmp(key, "ride_max_tiles") == 0) *value = ctx->ride_max_tiles;
  else if (std::strcmp(key, "ride_same_stream_max_tiles") == 0) *value = ctx->ride_same_stream_max_tiles;
  else LPGP_CHECK(false, "unknown option %s", key);
  return 0;
}

int lpgp_set_option(lpgp_ctx* ctx, const char* key, int64_t value) {
  LPGP_DEVICE(ctx);
  if (std::strcmp(key, "nb") == 0) {
    LPGP_CHECK(value >= TILE && value % TILE == 0, "nb must be a positive multiple of %d", TILE);
    ctx->nb = value;
  } else if (std::strcmp(key, "small_tiles_max") == 0) {
    ctx->small_tiles_max = (int)value;
  } else if (std::strcmp(key, "chain_us_tile") == 0) {
    ctx->chain_us_tile = (double)value;
  } else if (std::strcmp(key, "solve_chain_us_tile") == 0) {
    ctx->solve_chain_us_tile = (double)value;
  } else if (std::strcmp(key, "chain_us_fixed") == 0) {
    ctx->chain_us_fixed = (double)value;
  } else if (std::strcmp(key, "dense_tiles") == 0) {
    ctx->dense_tiles = (int)value;
  } else if (std::strcmp(key, "fused_solve") == 0) {
    ctx->fused_solve = (int)value;
  } else if (std::strcmp(key, "dist_bcast") == 0) {
    ctx->dist_bcast = value != 0;           // (the same on every rank)
  } else if (std::strcmp(key, "split_gather") == 0) {
    ctx->split_gather = value != 0;
  } else if (std::strcmp(key, "scoped_gather") == 0) {
    ctx->scoped_gather = value != 0;          // (the same on every rank)
  } else if (std::strcmp(key, "asm_factors") == 0) {
    ctx->asm_factors = value != 0;
  } else if (std::strcmp(key, "asm_fast") == 0) {
    ctx->asm_fast = value != 0;
  } else if (std::strcmp(key, "asm_batch") == 0) {
    ctx->asm_batch = value != 0;
  } else if (std::strcmp(key, "kron_wide") == 0) {
    ctx->kron_wide = value != 0;
  } else if (std::strcmp(key, "asm_ct") == 0) {
    LPGP_CHECK(value >= 1 && value <= 64, "asm_ct must be in 1 .. 64");
    ctx->asm_ct = (int)value;
  } else if (std::strcmp(key, "gemm3_fact") == 0) {
    ctx->gemm3_fact = value != 0;
  } else if (std::strcmp(key, "gemm3") == 0) {
    ctx->gemm3 = value < 0 ? lpgp_ctx().gemm3 : (int)value;       // (negative: back to the built-in default)
  } else if (std::strcmp(key, "small_ring2") == 0) {
    ctx->small_ring2 = (int)value;
  } else if (std::strcmp(key, "fused_ahead") == 0) {
    ctx->fused_ahead = (int)value;
  } else if (std::strcmp(key, "fused_ahead_min_us") == 0) {
    ctx->fused_ahead_min_us = (int)value;
  } else if (std::strcmp(key, "min_supertiles") == 0) {
    ctx->min_supertiles = (int)value;
  } else if (std::strcmp(key, "nb_solve") == 0) {
    LPGP_CHECK(value >= 0 && value % TILE == 0, "nb_solve must be a multiple of %d (0: nb)", TILE);
    ctx->nb_solve = value;
  } else if (std::strcmp(key, "nb_outer_solve") == 0) {
    LPGP_CHECK(value >= 0 && value % TILE == 0, "nb_outer_solve must be a multiple of %d (0 disables)", TILE);
    ctx->nb_outer_solve = value;
  } else if (std::strcmp(key, "nb_outer_solve_min_tiles") == 0) {
    LPGP_CHECK(value >= 0, "nb_outer_solve_min_tiles must be >= 0");
    ctx->nb_outer_solve_min_tiles = (int)value;
  } else if (std::strcmp(key, "nb_outer") == 0) {
    LPGP_CHECK(value >= 0 && value % TILE == 0, "nb_outer must be a multiple of %d (0 disables)", TILE);
    ctx->nb_outer = value;
  } else if (std::strcmp(key, "nb_outer_min_tiles") == 0) {
    ctx->nb_outer_min_tiles = (int)value;
  } else if (std::strcmp(key, "nb_big") == 0) {
    LPGP_CHECK(value >= 0 && value % TILE == 0, "nb_big must be a multiple of %d (0 disables)", TILE);
    ctx->nb_big = value;
  } else if (std::strcmp(key, "nb_big_min_tiles") == 0) {
    ctx->nb_big_min_tiles = (int)value;
  } else if (std::strcmp(key, "lookahead") == 0) {
    ctx->lookahead = value != 0;
  } else if (std::strcmp(key, "ride_stream") == 0) {
    ctx->ride_stream = (int)value;
  } else if (std::strcmp(key, "chain_resident_max_rows") == 0) {
    ctx->chain_resident_max_rows = (int)value;
  } else if (std::strcmp(key, "trsv_resident") == 0) {
    ctx->trsv_resident = value != 0;
  } else if (std::strcmp(key, "chain_resident2_max_rows") == 0) {
    ctx->chain_resident2_max_rows = (int)value;
  } else if (std::strcmp(key, "chain_ahead") == 0) {
    ctx->chain_ahead = value != 0;
  } else if (std::strcmp(key, "chain_ahead_min_rows") == 0) {
    ctx->chain_ahead_min_rows = (int)value;
  } else if (std::strcmp(key, "ride_vchain_max_wgs") == 0) {
    ctx->ride_vchain_max_wgs = (int)value;
  } else if (std::strcmp(key, "ride_occ3") == 0) {
    ctx->ride_occ3 = value != 0;
  } else if (std::strcmp(key, "ride_aug") == 0) {
    ctx->ride_aug = (int)value;
  } else if (std::strcmp(key, "append_split") == 0) {
    ctx->append_split = value != 0;
  } else if (std::strcmp(key, "ride_gate_pct") == 0) {
    ctx->ride_gate_pct = (int)value;
  } else if (std::strcmp(key, "ride_outer_rows") == 0) {
    LPGP_CHECK(value >= 0 && value % (4 * TILE) == 0, "ride_outer_rows must be a multiple of %d (0 disables)", 4 * TILE);
    ctx->ride_outer_rows = value;
  } else if (std::strcmp(key, "ride_outer_min_tiles") == 0) {
    ctx->ride_outer_min_tiles = (int)value;
  } else if (std::strcmp(key, "ride_max_tiles") == 0) {
    ctx->ride_max_tiles = (int)value;
  } else if (std::strcmp(key, "ride_same_stream_max_tiles") == 0) {
    ctx->ride_same_stream_max_tiles = (int)value;
  } else {
    LPGP_CHECK(false, "unknown option %s", key);
  }
  return 0;
}

// ---- multi-GPU ---------------------------------------------------------------------------
int lpgp_dist_unique_id(char* out128) {
  LPGP_CHECK(out128 != nullptr, "lpgp_dist_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  LPGP_CHECK(r == ncclSuccess, "ncclGetUniqueId: %s", ncclGetErrorString(r));
  std::memcpy(out128, &id, 128);
  return 0;
}

// default process grid for `world` ranks: Pr = world, Pc = 1.  On the full mesh of xGMI links (every GPU a direct
// link to every other) a gathered panel of S bytes costs a link S / Pr: a tall grid keeps
// every link at 1 / world of the panel, shares the panel triangular solve among all ranks and needs no row exchange.
static void choose_grid(lpgp_ctx* ctx, int world) {
  if (ctx->grid_set && ctx->pr * ctx->pc == world) return;
  ctx->pr = world;
  ctx->pc = 1;
}

static int dist_fail(lpgp_ctx* ctx, int rc);

int lpgp_dist_set_grid(lpgp_ctx* ctx, int32_t pr, int32_t pc) {
  LPGP_CHECK(ctx && pr >= 1 && pc >= 1 && pr <= 8 && pc <= 8, "lpgp_dist_set_grid: bad grid %d x %d", pr, pc);
  // before the bring-up, or afterwards while no matrix or right-hand side exists (their storage is laid out for the
  // grid they were created under): bench.py times the same workload on two grids during warm-up and keeps the faster
  if (ctx->distributed()) {
    LPGP_CHECK(pr * pc == ctx->world, "lpgp_dist_set_grid: grid %d x %d does not match %d ranks", pr, pc, ctx->world);
    LPGP_CHECK(ctx->live_mats == 0, "lpgp_dist_set_grid: %d matrices / right-hand sides of the current grid are still alive", ctx->live_mats);
  }
  ctx->pr = pr;
  ctx->pc = pc;
  ctx->grid_set = 1;
  return 0;
}

int lpgp_dist_link_probe(lpgp_ctx* ctx, int64_t bytes, int32_t reps, double* out) {
  LPGP_CHECK(ctx && out && bytes >= 8 && reps >= 1, "lpgp_dist_link_probe: bad argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(ctx->distributed() && !ctx->dist_broken, "lpgp_dist_link_probe: no multi-GPU job (call lpgp_dist_init first)");
  return dist_fail(ctx, dist_link_probe(ctx, bytes, reps, out));
}

int lpgp_dist_grid(lpgp_ctx* ctx, int32_t* pr, int32_t* pc) {
  LPGP_CHECK(ctx != nullptr, "lpgp_dist_grid: null context");
  if (pr) *pr = ctx->pr;
  if (pc) *pc = ctx->pc;
  return 0;
}

int lpgp_dist_stats(lpgp_ctx* ctx, double* bytes_sent, double* bytes_received, int32_t reset) {
  LPGP_CHECK(ctx != nullptr, "lpgp_dist_stats: null context");
  if (bytes_sent) *bytes_sent = ctx->comm_bytes_sent;
  if (bytes_received) *bytes_received = ctx->comm_bytes_recv;
  if (reset) ctx->comm_bytes_sent = ctx->comm_bytes_recv = 0.0;
  return 0;
}

int lpgp_dist_init(lpgp_ctx* ctx, int32_t rank, int32_t world, const char* uid128) {
  LPGP_CHECK(ctx && uid128 && world >= 1 && rank >= 0 && rank < world, "lpgp_dist_init: bad argument");
  LPGP_CHECK(!ctx->distributed(), "lpgp_dist_init: already initialised");
  LPGP_CHECK(!ctx->grid_set || ctx->pr * ctx->pc == world, "lpgp_dist_init: grid %d x %d does not match %d ranks", ctx->pr, ctx->pc, world);
  LPGP_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, uid128, 128);
  ncclComm_t comm = nullptr;
  ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
  LPGP_CHECK(r == ncclSuccess, "ncclCommInitRank(rank %d of %d): %s", rank, world, ncclGetErrorString(r));
  ctx->nccl_comm = comm;
  ctx->rank = rank;
  ctx->world = world;
  choose_grid(ctx, world);
  if (world > 1 && ctx->split_gather) {
    // a second communicator for the bulk of split panel gathers (operations on one communicator are serialised: the
    // small exchanges of the panel chain would queue behind it).  Without it the split gather still works, serialised.
    ncclComm_t bulk = nullptr;
    if (ncclCommSplit(comm, 0, rank, &bulk, nullptr) == ncclSuccess) ctx->nccl_comm_bulk = bulk;
  }
  // every connection is made NOW, while all ranks are alive (see dist_warm_up, which also makes the ranks AGREE on the
  // bulk communicator: all of them use it or none does); also the first check that data arrives
  int rc = dist_fail(ctx, dist_warm_up(ctx));
  if (rc != 0) {
    // a failed bring-up leaves a plain single-GPU context behind (the caller may try another transport, or run replicas)
    ctx->rank = 0;
    ctx->world = 1;
    if (!ctx->grid_set) ctx->pr = ctx->pc = 1;
    ctx->dist_broken = 0;
  }
  return rc;
}

int lpgp_dist_init_host(lpgp_ctx* ctx, int32_t rank, int32_t world, lpgp_host_exchange_fn fn, void* user) {
  LPGP_CHECK(ctx && fn && world >= 1 && rank >= 0 && rank < world, "lpgp_dist_init_host: bad argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(!ctx->distributed(), "lpgp_dist_init_host: already initialised");
  LPGP_CHECK(!ctx->grid_set || ctx->pr * ctx->pc == world, "lpgp_dist_init_host: grid %d x %d does not match %d ranks", ctx->pr, ctx->pc, world);
  ctx->host_xfer = fn;
  ctx->host_xfer_user = user;
  ctx->rank = rank;
  ctx->world = world;
  choose_grid(ctx, world);
  return 0;
}

int lpgp_dist_ipc_export(lpgp_ctx* ctx, int64_t window_bytes, char* handle64) {
  LPGP_CHECK(ctx && handle64 && window_bytes >= (int64_t)(1 << 20), "lpgp_dist_ipc_export: bad argument (window of at least 1 MiB)");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(!ctx->distributed() && !ctx->ipc_window, "lpgp_dist_ipc_export: already initialised");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t size");
  const size_t doubles = (size_t)window_bytes / sizeof(double);
  LPGP_HIP(hipMalloc(&ctx->ipc_window, doubles * sizeof(double)));
  ctx->ipc_window_doubles = doubles;
  hipIpcMemHandle_t h;
  LPGP_HIP(hipIpcGetMemHandle(&h, ctx->ipc_window));
  std::memcpy(handle64, &h, 64);
  return 0;
}

int lpgp_dist_init_ipc(lpgp_ctx* ctx, int32_t rank, int32_t world, const char* handles, lpgp_host_exchange_fn fn, void* user) {
  LPGP_CHECK(ctx && handles && fn && world >= 1 && rank >= 0 && rank < world, "lpgp_dist_init_ipc: bad argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(!ctx->distributed() && ctx->ipc_window, "lpgp_dist_init_ipc: call lpgp_dist_ipc_export first (once)");
  LPGP_CHECK(!ctx->grid_set || ctx->pr * ctx->pc == world, "lpgp_dist_init_ipc: grid %d x %d does not match %d ranks", ctx->pr, ctx->pc, world);
  ctx->ipc_peer.assign((size_t)world, nullptr);
  for (int r = 0; r < world; ++r) {
    if (r == rank) {
      ctx->ipc_peer[r] = ctx->ipc_window;
      continue;
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, handles + (size_t)r * 64, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      set_error("lpgp_dist_init_ipc: hipIpcOpenMemHandle for the window of rank %d: %s", r, hipGetErrorString(e));
      ctx->ipc_peer.clear();
      return -1;
    }
    ctx->ipc_peer[r] = (double*)p;
  }
  ctx->host_xfer = fn;                 // barriers and the info all-reduce travel over the caller's control plane
  ctx->host_xfer_user = user;
  ctx->rank = rank;
  ctx->world = world;
  choose_grid(ctx, world);
  return 0;
}

int lpgp_dist_info(lpgp_ctx* ctx, int32_t* rank, int32_t* world) {
  LPGP_DEVICE(ctx);
  if (rank) *rank = ctx->rank;
  if (world) *world = ctx->world;
  return 0;
}

// ---- points ----------------------------------------------------------------------------
int lpgp_pts_create(lpgp_ctx* ctx, const double* X_host, int64_t n, int32_t d, lpgp_pts** out) {
  LPGP_CHECK(ctx && X_host && out, "lpgp_pts_create: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(n >= 0 && d >= 1 && d <= LPGP_MAXD, "lpgp_pts_create: n=%lld d=%d", (long long)n, d);
  lpgp_pts* p = new lpgp_pts();
  p->ctx = ctx;
  p->n = n;
  p->d = d;
  p->n_pad = round_up(n > 0 ? n : 1, 64);
  const size_t bytes = (size_t)p->n_pad * d * sizeof(double);
  const bool recycle = !ctx->distributed() && !ctx->single_stream;
  // the buffer: a recycled one of about the size (lpgp_internal.h: pts_pool), else a new one
  p->x = nullptr;
  if (recycle) {
    int best = -1;
    for (int i = 0; i < (int)ctx->pts_pool.size(); ++i) {
      const auto& b = ctx->pts_pool[i];
      if (b.bytes >= bytes && b.bytes <= 2 * bytes + 4096 && (best < 0 || b.bytes < ctx->pts_pool[best].bytes)) best = i;
    }
    if (best >= 0) {
      p->x = (double*)ctx->pts_pool[best].p;
      p->bytes = ctx->pts_pool[best].bytes;
      ctx->pts_pool.erase(ctx->pts_pool.begin() + best);
    }
  }
  if (!p->x) {
    hipError_t e = hipMalloc(&p->x, bytes);
    if (e != hipSuccess) {
      set_error("lpgp_pts_create: device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
      delete p;
      return -1;
    }
    p->bytes = bytes;
  }
  auto fill = [&](double* soa) {
    std::memset(soa, 0, bytes);
    for (int64_t i = 0; i < n; ++i)
      for (int j = 0; j < d; ++j) soa[(size_t)j * p->n_pad + i] = X_host[i * d + j];
  };
  hipError_t e = hipSuccess;
  if (recycle && bytes <= lpgp_ctx::PTS_SLOT_BYTES) {
    // small: through the next pinned slot, on the panel stream, no wait (the slot is reused after its copy has completed)
    lpgp_ctx::PtsSlot& sl = ctx->pts_ring[ctx->pts_next];
    ctx->pts_next = (ctx->pts_next + 1) % lpgp_ctx::PTS_SLOTS;
    if (!sl.h) {
      e = hipHostMalloc(&sl.h, lpgp_ctx::PTS_SLOT_BYTES, hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
    } else if (sl.used) {
      e = hipEventSynchronize(sl.done);
    }
    if (e == hipSuccess) {
      fill((double*)sl.h);
      e = hipMemcpyAsync(p->x, sl.h, bytes, hipMemcpyHostToDevice, ctx->s_main);
    }
    if (e == hipSuccess) e = hipEventRecord(sl.done, ctx->s_main);
    if (e == hipSuccess) sl.used = true;
  } else {
    std::vector<double> soa((size_t)p->n_pad * d);
    fill(soa.data());
    // (a recycled buffer may still be read by kernels on the panel stream: the copy is ordered behind them there)
    e = hipMemcpyAsync(p->x, soa.data(), bytes, hipMemcpyHostToDevice, ctx->s_main);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->s_main);
  }
  if (e != hipSuccess) {
    set_error("lpgp_pts_create: upload: %s", hipGetErrorString(e));
    (void)hipFree(p->x);
    delete p;
    return -1;
  }
  *out = p;
  return 0;
}

int lpgp_pts_destroy(lpgp_pts* p) {
  if (!p) return 0;
  lpgp_ctx* ctx = p->ctx;
  (void)hipSetDevice(ctx->device);
  // recycled: whatever still reads it runs on the panel stream, where the next owner's upload is ordered behind it
  if (!ctx->distributed() && !ctx->single_stream && ctx->pts_pool.size() < 32 && p->bytes <= ((size_t)4 << 20))
    ctx->pts_pool.push_back({p->x, p->bytes});
  else
    (void)hipFree(p->x);
  delete p;
  return 0;
}

// ---- matrix ----------------------------------------------------------------------------
int lpgp_mat_create(lpgp_ctx* ctx, int64_t capacity_hint, lpgp_mat** out) {
  LPGP_CHECK(ctx && out, "lpgp_mat_create: null argument");
  LPGP_DEVICE(ctx);
  lpgp_mat* m = new lpgp_mat();
  m->ctx = ctx;
  m->cap = 0;
  m->a = m->linv = m->w = nullptr;
  m->n = m->pn = m->pn_fact = 0;
  m->has_w = 0;
  m->has_r = 0;
  int64_t cap = round_up(capacity_hint > 0 ? capacity_hint : 8 * TILE, TILE);      // (no hint: room for 1 024 rows, 8 MB -- the reference's own problem sizes start here)
  int rc = mat_alloc(ctx, m, cap);
  if (rc != 0) {
    delete m;
    return rc;
  }
  ++ctx->live_mats;
  *out = m;
  return 0;
}

int lpgp_mat_destroy(lpgp_mat* m) {
  if (!m) return 0;
  (void)hipSetDevice(m->ctx->device);
  mat_release(m->ctx, m);
  --m->ctx->live_mats;
  delete m;
  return 0;
}

static int mat_add_block_impl(lpgp_ctx* ctx, lpgp_mat* mat, int64_t n, bool pad_now);
int lpgp_mat_add_block(lpgp_ctx* ctx, lpgp_mat* mat, int64_t n) { return mat_add_block_impl(ctx, mat, n, true); }
// pad_now = false: the caller launches the identity tail itself (lpgp_mat_condition: together with the noise, launch_finish_block)
static int mat_add_block_impl(lpgp_ctx* ctx, lpgp_mat* mat, int64_t n, bool pad_now) {
  LPGP_CHECK(ctx && mat && n > 0, "lpgp_mat_add_block: bad argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_mat_add_block");
  LPGP_CHECK(mat->hidden.empty(), "lpgp_mat_add_block: a strict prefix of the blocks is in view (lpgp_mat_set_view); extend a clone instead");
  lpgp_block b;
  b.n = n;
  b.off = mat->n;
  b.poff = mat->pn;
  b.pn = round_up(n, TILE);
  if (b.poff + b.pn > mat->cap) {
    // grow geometrically while the matrix is small (a chain of small conditionings without `gram_capacity_hint` used to
    // re-allocate, copy and synchronise at EVERY block: 48 us of host time per add_block at N_tot = 1 152); a block that
    // needs more than that gets exactly what it needs (the large PDE block of c3 / c4 is the last growth of its chain)
    int64_t want = b.poff + b.pn;
    const int64_t geo = round_up(mat->cap + mat->cap / 2, TILE);
    // ... capped by BYTES, not rows (a chain that needed just over 10 923 rows got a 16 384^2 matrix, 2.1 GB and
    // a full copy, about twice what it asked for): at most 256 MB of storage beyond the request
    if (geo > want && (geo * geo - want * want) * (int64_t)sizeof(double) <= ((int64_t)256 << 20)) want = geo;
    int rc = mat_alloc(ctx, mat, want);
    if (rc != 0) return rc;
  }
  const int64_t pad = b.pn - b.n;
  if (pad > 0 && pad_now) {
    // identity tail of the block: zero the pad rows (all columns up to the block end) and the
    // pad columns (all rows down to the capacity), then ones on the diagonal
    const Layout2D lay = mat_layout(ctx);
    const int64_t r0 = b.poff + b.n;
    launch_pad_block(ctx->s_main, mat->a, mat->lr_cap, r0, pad, mat->cap, lay);
    LPGP_HIP(hipGetLastError());
  }
  mat->blocks.push_back(b);
  mat->n += n;
  mat->pn += b.pn;
  mat->has_w = 0;
  mat->has_r = 0;
  return (int)mat->blocks.size() - 1;
}

int lpgp_mat_pop_block(lpgp_ctx* ctx, lpgp_mat* mat) {
  LPGP_CHECK(ctx && mat, "lpgp_mat_pop_block: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(mat->hidden.empty(), "lpgp_mat_pop_block: a strict prefix of the blocks is in view");
  LPGP_CHECK(!mat->blocks.empty(), "lpgp_mat_pop_block: no block");
  const lpgp_block b = mat->blocks.back();
  LPGP_CHECK(b.poff >= mat->pn_fact, "lpgp_mat_pop_block: the last block is part of the factor");
  LPGP_TRY(sync_stream(ctx, ctx->s_main));          // nothing of a failed append still in flight (watched: an exchange with a dead peer may be pending)
  mat->blocks.pop_back();
  mat->n -= b.n;
  mat->pn -= b.pn;
  mat->has_w = 0;
  mat->has_r = 0;
  return 0;
}

int32_t lpgp_mat_num_blocks(const lpgp_mat* mat) { return mat ? (int32_t)mat->blocks.size() : -1; }
int32_t lpgp_mat_num_blocks_total(const lpgp_mat* mat) { return mat ? (int32_t)(mat->blocks.size() + mat->hidden.size()) : -1; }

int lpgp_mat_set_view(lpgp_ctx* ctx, lpgp_mat* mat, int32_t nblocks) {
  LPGP_CHECK(ctx && mat, "lpgp_mat_set_view: null argument");
  const int total = (int)(mat->blocks.size() + mat->hidden.size());
  if (nblocks < 0) nblocks = total;
  LPGP_CHECK(nblocks >= 1 && nblocks <= total, "lpgp_mat_set_view: %d of %d blocks", nblocks, total);
  if (nblocks == (int)mat->blocks.size()) return 0;
  const int64_t fact_all = mat->hidden.empty() ? mat->pn_fact : mat->pn_fact_all;
  std::vector<lpgp_block> all = mat->blocks;
  all.insert(all.end(), mat->hidden.begin(), mat->hidden.end());
  const lpgp_block& last = all[nblocks - 1];
  LPGP_CHECK(nblocks == total || last.poff + last.pn <= fact_all, "lpgp_mat_set_view: block %d is not factored yet", nblocks - 1);
  mat->blocks.assign(all.begin(), all.begin() + nblocks);
  mat->hidden.assign(all.begin() + nblocks, all.end());
  mat->n = last.off + last.n;
  mat->pn = last.poff + last.pn;
  mat->pn_fact_all = fact_all;
  mat->pn_fact = fact_all < mat->pn ? fact_all : mat->pn;
  mat->has_w = 0;                                    // resident weights / residual belong to the previous view
  mat->has_r = 0;
  return 0;
}

int lpgp_mat_clone(lpgp_ctx* ctx, const lpgp_mat* src, int32_t nblocks, lpgp_mat** out) {
  LPGP_CHECK(ctx && src && out, "lpgp_mat_clone: null argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(src, "lpgp_mat_clone");
  LPGP_CHECK(!ctx->distributed(), "lpgp_mat_clone: not available in a multi-GPU job");
  std::vector<lpgp_block> all = src->blocks;
  all.insert(all.end(), src->hidden.begin(), src->hidden.end());
  LPGP_CHECK(nblocks >= 1 && nblocks <= (int)all.size(), "lpgp_mat_clone: %d of %d blocks", nblocks, (int)all.size());
  const int64_t fact_all = src->hidden.empty() ? src->pn_fact : src->pn_fact_all;
  const lpgp_block& last = all[nblocks - 1];
  const int64_t pn = last.poff + last.pn;
  LPGP_CHECK(pn <= fact_all, "lpgp_mat_clone: block %d is not factored yet", nblocks - 1);
  lpgp_mat* m = nullptr;
  int rc = lpgp_mat_create(ctx, pn, &m);
  if (rc != 0) return rc;
  hipError_t e = hipMemcpy2DAsync(m->a, (size_t)m->lr_cap * sizeof(double), src->a, (size_t)src->lr_cap * sizeof(double),
                                  (size_t)pn * sizeof(double), (size_t)pn, hipMemcpyDeviceToDevice, ctx->s_main);
  if (e == hipSuccess)
    e = hipMemcpyAsync(m->linv, src->linv, (size_t)pn * TILE * sizeof(double), hipMemcpyDeviceToDevice, ctx->s_main);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->s_main);
  if (e != hipSuccess) {
    set_error("lpgp_mat_clone: %s", hipGetErrorString(e));
    lpgp_mat_destroy(m);
    return -1;
  }
  m->blocks.assign(all.begin(), all.begin() + nblocks);
  m->n = last.off + last.n;
  m->pn = pn;
  m->pn_fact = pn;
  *out = m;
  return 0;
}

int64_t lpgp_mat_size(const lpgp_mat* mat) { return mat ? mat->n : -1; }
int64_t lpgp_mat_padded_size(const lpgp_mat* mat) { return mat ? mat->pn : -1; }

int lpgp_gram_assemble(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups, const lpgp_pts* X0,
                       const lpgp_pts* X1, lpgp_mat* mat, int32_t bi, int32_t bj) {
  LPGP_CHECK(ctx && kd && X0 && mat, "lpgp_gram_assemble: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(bi >= 0 && bi < (int)mat->blocks.size() && bj >= 0 && bj <= bi, "lpgp_gram_assemble: bad block (%d,%d)", bi, bj);
  const lpgp_block& Bi = mat->blocks[bi];
  const lpgp_block& Bj = mat->blocks[bj];
  LPGP_CHECK(Bi.poff >= mat->pn_fact, "lpgp_gram_assemble: block %d is already factored", bi);
  LPGP_CHECK(X0->n == Bi.n, "lpgp_gram_assemble: X0 has %lld points, block %d has %lld rows", (long long)X0->n, bi, (long long)Bi.n);
  const bool sym = (X1 == nullptr);
  LPGP_CHECK(sym == (bi == bj), "lpgp_gram_assemble: X1 must be NULL exactly for diagonal blocks");
  const lpgp_pts* Xc = sym ? X0 : X1;
  LPGP_CHECK(Xc->n == Bj.n && Xc->d == X0->d && kd[0].d == X0->d, "lpgp_gram_assemble: shape mismatch");
  DevDesc desc;
  int rc = lower_kdesc(kd, ngroups, &desc);
  if (rc != 0) return rc;
  // distributed factorisation: a rank assembles only the columns of the panels it will factor
  // (the others arrive as factored panels); columns that are already factored (cross blocks of
  // a block append) are needed by every rank
  // multi-GPU: a rank evaluates exactly the tiles it owns (2-D block-cyclic), zero communication
  rc = launch_assemble(ctx, ctx->s_main, desc, X0->x, X0->n, X0->n_pad, Xc->x, Xc->n, Xc->n_pad, mat->a, mat->lr_cap,
                       Bi.poff, Bj.poff, sym ? 1 : 0, mat_layout(ctx));
  return rc;       // asynchronous: consumers are ordered behind it on the main stream
}

int lpgp_kron_fits(const lpgp_kdesc* kd, int32_t ngroups) { return kron_fits(kd, ngroups) ? 1 : 0; }

int lpgp_gram_assemble_grid(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups, const lpgp_pts* const* F0,
                            const lpgp_pts* const* F1, lpgp_mat* mat, int32_t bi, int32_t bj) {
  LPGP_CHECK(ctx && kd && F0 && mat, "lpgp_gram_assemble_grid: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(ngroups >= 1 && ngroups <= LPGP_MAXG, "lpgp_gram_assemble_grid: bad ngroups %d", ngroups);
  LPGP_CHECK(bi >= 0 && bi < (int)mat->blocks.size() && bj >= 0 && bj <= bi, "lpgp_gram_assemble_grid: bad block (%d,%d)", bi, bj);
  const lpgp_block& Bi = mat->blocks[bi];
  const lpgp_block& Bj = mat->blocks[bj];
  LPGP_CHECK(Bi.poff >= mat->pn_fact, "lpgp_gram_assemble_grid: block %d is already factored", bi);
  const bool sym = (F1 == nullptr);
  LPGP_CHECK(sym == (bi == bj), "lpgp_gram_assemble_grid: F1 must be NULL exactly for diagonal blocks");
  const int D = kd[0].d;
  LPGP_CHECK(D >= 1 && D <= LPGP_MAXD, "lpgp_gram_assemble_grid: d=%d", D);
  const double *f0[LPGP_MAXD], *f1[LPGP_MAXD];
  int64_t n0d[LPGP_MAXD], n1d[LPGP_MAXD], n0 = 1, n1 = 1;
  for (int d = 0; d < D; ++d) {
    const lpgp_pts* a = F0[d];
    const lpgp_pts* b = sym ? F0[d] : F1[d];
    LPGP_CHECK(a && b && a->d == 1 && b->d == 1 && a->n >= 1 && b->n >= 1, "lpgp_gram_assemble_grid: factor %d must be a 1-D point set", d);
    f0[d] = a->x; f1[d] = b->x;
    n0d[d] = a->n; n1d[d] = b->n;
    n0 *= a->n; n1 *= b->n;
  }
  LPGP_CHECK(n0 == Bi.n && n1 == Bj.n, "lpgp_gram_assemble_grid: grids have %lld x %lld points, block (%d,%d) is %lld x %lld",
             (long long)n0, (long long)n1, bi, bj, (long long)Bi.n, (long long)Bj.n);
  const size_t wd = kron_work_doubles(D, n0d, n1d);
  void* work = nullptr;
  if (pool_alloc(ctx, &work, wd * sizeof(double), nullptr) != 0) return -1;
  int rc = launch_assemble_kron(ctx, ctx->s_main, kd, ngroups, f0, n0d, f1, n1d, (double*)work, wd, mat->a, mat->lr_cap,
                                Bi.poff, Bj.poff, sym ? 1 : 0, mat_layout(ctx));
  pool_free(ctx, work, wd * sizeof(double));       // reuse is ordered behind this launch on the main stream
  return rc;
}

int lpgp_mat_add_diag(lpgp_ctx* ctx, lpgp_mat* mat, int32_t bi, const double* v_host, double scalar) {
  LPGP_CHECK(ctx && mat && bi >= 0 && bi < (int)mat->blocks.size(), "lpgp_mat_add_diag: bad argument");
  LPGP_DEVICE(ctx);
  const lpgp_block& B = mat->blocks[bi];
  LPGP_CHECK(B.poff >= mat->pn_fact, "lpgp_mat_add_diag: block %d is already factored", bi);
  double* dv = nullptr;
  if (v_host) {
    int rc = ensure_tmp(ctx, B.n);
    if (rc != 0) return rc;
    LPGP_HIP(hipMemcpyAsync(ctx->d_tmp, v_host, (size_t)B.n * sizeof(double), hipMemcpyHostToDevice, ctx->s_main));
    dv = ctx->d_tmp;
  }
  int rc = launch_add_diag(ctx->s_main, mat->a, mat->lr_cap, B.poff, B.n, dv, scalar, mat_layout(ctx));
  if (rc != 0) return rc;
  if (v_host) LPGP_HIP(hipStreamSynchronize(ctx->s_main));     // borrowed host vector, shared scratch
  return 0;
}

int lpgp_mat_add_dense(lpgp_ctx* ctx, lpgp_mat* mat, int32_t bi, const double* B_host) {
  LPGP_CHECK(ctx && mat && B_host && bi >= 0 && bi < (int)mat->blocks.size(), "lpgp_mat_add_dense: bad argument");
  LPGP_DEVICE(ctx);
  const lpgp_block& B = mat->blocks[bi];
  LPGP_CHECK(B.poff >= mat->pn_fact, "lpgp_mat_add_dense: block %d is already factored", bi);
  int rc = ensure_tmp(ctx, B.n * B.n);
  if (rc != 0) return rc;
  LPGP_HIP(hipMemcpyAsync(ctx->d_tmp, B_host, (size_t)(B.n * B.n) * sizeof(double), hipMemcpyHostToDevice, ctx->s_main));
  rc = launch_add_dense(ctx->s_main, mat->a, mat->lr_cap, B.poff, B.n, ctx->d_tmp, mat_layout(ctx));
  if (rc != 0) return rc;
  LPGP_HIP(hipStreamSynchronize(ctx->s_main));
  return 0;
}

// A failure inside a collective call on ONE rank (HIP error, failed allocation, RCCL error) would leave the other
// ranks blocked in their next send / receive for ever: abort the communicator, which makes every pending and later
// RCCL call on every rank of the job return an error instead.  The context is unusable afterwards.
static int dist_fail(lpgp_ctx* ctx, int rc) {
  if (rc != 0 && ctx->nccl_comm && ctx->world > 1) {
    (void)ncclCommAbort((ncclComm_t)ctx->nccl_comm);
    if (ctx->nccl_comm_bulk) (void)ncclCommAbort((ncclComm_t)ctx->nccl_comm_bulk);
    ctx->nccl_comm = ctx->nccl_comm_bulk = nullptr;
    ctx->dist_broken = 1;
  }
  return rc;
}

int lpgp_mat_to_host(lpgp_ctx* ctx, lpgp_mat* mat, int32_t what, double* out_host) {
  LPGP_CHECK(ctx && mat && out_host, "lpgp_mat_to_host: null argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_mat_to_host");
  LPGP_CHECK(what == 0 || what == 1, "lpgp_mat_to_host: what must be 0 or 1");
  if (what == 0) LPGP_CHECK(mat->pn_fact == 0, "lpgp_mat_to_host: Gram no longer available after potrf");
  if (what == 1) LPGP_CHECK(mat->pn_fact == mat->pn, "lpgp_mat_to_host: matrix is not (fully) factored");
  const int64_t pn = mat->pn, n = mat->n;
  std::vector<double> tmp((size_t)pn * pn);
  LPGP_TRY(sync_stream(ctx, ctx->s_main));           // assembly launches are asynchronous
  if (ctx->distributed()) {
    LPGP_CHECK(what == 1, "lpgp_mat_to_host: only the factor can be collected in a multi-GPU job");
    int rc = dist_fail(ctx, factor_to_host_dist(ctx, mat, tmp.data()));      // collective: the factor is streamed to every rank
    if (rc != 0) return rc;
  } else
  LPGP_HIP(hipMemcpy2D(tmp.data(), (size_t)pn * sizeof(double), mat->a, (size_t)mat->lr_cap * sizeof(double),
                       (size_t)pn * sizeof(double), (size_t)pn, hipMemcpyDeviceToHost));
  // tmp is column-major pn x pn: element (r,c) at tmp[r + c*pn]
  for (const auto& bi : mat->blocks)
    for (const auto& bj : mat->blocks)
      for (int64_t i = 0; i < bi.n; ++i)
        for (int64_t j = 0; j < bj.n; ++j) {
          const int64_t r = bi.poff + i, c = bj.poff + j;
          double v;
          if (r >= c) v = tmp[(size_t)(r + c * pn)];
          else v = (what == 0) ? tmp[(size_t)(c + r * pn)] : 0.0;
          out_host[(bi.off + i) * n + (bj.off + j)] = v;
        }
  return 0;
}

int lpgp_mat_factor_diag(lpgp_ctx* ctx, lpgp_mat* mat, double* out_host) {
  LPGP_CHECK(ctx && mat && out_host, "lpgp_mat_factor_diag: null argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_mat_factor_diag");
  LPGP_CHECK(mat->pn_fact == mat->pn, "lpgp_mat_factor_diag: matrix is not (fully) factored");
  const int64_t pn = mat->pn;
  if (pn == 0) return 0;
  std::vector<double> tmp((size_t)pn);
  LPGP_TRY(sync_stream(ctx, ctx->s_main));
  if (ctx->distributed()) {
    // replicated diagonal blocks: block K at dblk + K nb^2, leading dimension nb
    const int64_t nb = ctx->nb;
    for (int64_t k0 = 0; k0 < pn; k0 += nb) {
      const int64_t h = std::min(nb, pn - k0);
      LPGP_HIP(hipMemcpy2D(tmp.data() + k0, sizeof(double), mat->dblk + (k0 / nb) * nb * nb, (size_t)(nb + 1) * sizeof(double),
                           sizeof(double), (size_t)h, hipMemcpyDeviceToHost));
    }
  } else {
    LPGP_HIP(hipMemcpy2D(tmp.data(), sizeof(double), mat->a, (size_t)(mat->lr_cap + 1) * sizeof(double), sizeof(double), (size_t)pn,
                         hipMemcpyDeviceToHost));
  }
  for (const auto& b : mat->blocks)
    for (int64_t i = 0; i < b.n; ++i) out_host[b.off + i] = tmp[(size_t)(b.poff + i)];
  return 0;
}

// ---- factor + solve -----------------------------------------------------------------------
int lpgp_potrf(lpgp_ctx* ctx, lpgp_mat* mat, int32_t* info) {
  LPGP_CHECK(ctx && mat, "lpgp_potrf: null argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_potrf");
  if (info) *info = 0;
  if (mat->pn_fact == mat->pn) return 0;
  LPGP_CHECK(mat->hidden.empty(), "lpgp_potrf: a strict prefix of the blocks is in view");
  int32_t h = 0;
  LPGP_CHECK(!ctx->dist_broken, "lpgp_potrf: the communicator was aborted after an earlier failure");
  int rc = ctx->distributed()
               ? dist_fail(ctx, potrf_dist(ctx, mat, mat->pn_fact / TILE, mat->pn / TILE, &h))
               : potrf_blocked(ctx, mat, mat->pn_fact / TILE, mat->pn / TILE, &h);
  if (rc != 0) {
    mat->poisoned = 1;               // (the matrix is partly overwritten in place, pn_fact unchanged: not a state to factor again from)
    return rc;
  }
  if (info) *info = h;
  if (h == 0) mat->pn_fact = mat->pn;
  mat->has_w = 0;
  mat->has_r = 0;
  return 0;
}

int lpgp_potrf_enqueue(lpgp_ctx* ctx, lpgp_mat* mat) {
  LPGP_CHECK(ctx && mat, "lpgp_potrf_enqueue: null argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_potrf_enqueue");
  if (mat->pn_fact == mat->pn) return 0;
  LPGP_CHECK(mat->hidden.empty(), "lpgp_potrf_enqueue: a strict prefix of the blocks is in view");
  LPGP_CHECK(!ctx->distributed(), "lpgp_potrf_enqueue: single GPU only (the multi-GPU factorisation agrees on its status collectively: lpgp_potrf)");
  int rc = potrf_blocked(ctx, mat, mat->pn_fact / TILE, mat->pn / TILE, nullptr);
  if (rc != 0) {
    mat->poisoned = 1;
    return rc;
  }
  mat->pn_fact = mat->pn;          // provisionally: lpgp_mat_check / lpgp_mat_truncate take it back on failure
  mat->unchecked = 1;
  mat->status_known = 0;
  mat->has_w = 0;
  mat->has_r = 0;
  return 0;
}

int lpgp_mat_condition(lpgp_ctx* ctx, lpgp_mat* mat, int64_t n, const lpgp_pts* X_new, const lpgp_cond_block* row, int32_t nrow,
                       double noise_scalar, const double* noise_diag, const double* noise_dense, int32_t lazy, int32_t* info) {
  LPGP_CHECK(ctx && mat && row && n > 0, "lpgp_mat_condition: bad argument");
  LPGP_CHECK(nrow == (int32_t)mat->blocks.size() + 1, "lpgp_mat_condition: %d row entries for %d earlier blocks", nrow, (int)mat->blocks.size());
  LPGP_CHECK((noise_diag != nullptr) + (noise_dense != nullptr) + (noise_scalar != 0.0) <= 1, "lpgp_mat_condition: more than one form of noise");
  LPGP_CHECK(lazy == 0 || !ctx->distributed(), "lpgp_mat_condition: lazy status on a single GPU only");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_mat_condition");
  if (info) *info = 0;
  // earlier blocks that were only assembled (lazy == 2): another deferred block joins them -- they are factored TOGETHER by
  // whoever needs the factor first, one factorisation from the first unfactored column on (lpgp_potrf_predict: with the
  // prediction riding inside all of it); lazy == 1 enqueues their factorisation first, lazy == 0 requires them factored
  if (lazy == 1 && mat->pn_fact < mat->pn) LPGP_TRY(lpgp_potrf_enqueue(ctx, mat));
  LPGP_CHECK(lazy == 2 || mat->pn_fact == mat->pn, "lpgp_mat_condition: earlier blocks are not factored (lpgp_potrf first)");
  // (the identity tail of the block is written by ONE launch together with its noise, behind the assembly -- fused_finish;
  //  the assembly kernels touch logical entries only)
  const bool fused_finish = ctx->asm_batch && !noise_dense;
  const int bi = mat_add_block_impl(ctx, mat, n, !fused_finish);
  if (bi < 0) return bi;
  int rc = 0;
  if (ctx->asm_batch && !ctx->distributed()) {
    // the block row with as few launches as it has DIFFERENT descriptors: consecutive entries of the per-entry path that
    // share one (value observations against value observations: the whole row incl. the diagonal block; the cross blocks
    // of a differential block against the boundary blocks) go into one launch (assemble.hip: launch_assemble_batch)
    const lpgp_block Bi = mat->blocks[bi];
    std::vector<DevDesc> descs((size_t)nrow);
    std::vector<AsmJob> jobs;
    int run_start = -1;
    auto flush = [&](int upto) -> int {
      if (run_start < 0 || jobs.empty()) { run_start = -1; jobs.clear(); return 0; }
      int r = launch_assemble_batch(ctx, ctx->s_main, descs[(size_t)run_start], jobs.data(), (int)jobs.size(), mat->a, mat->lr_cap, mat_layout(ctx));
      (void)upto;
      run_start = -1;
      jobs.clear();
      return r;
    };
    for (int j = 0; j < nrow && rc == 0; ++j) {
      const lpgp_cond_block& e = row[j];
      if (e.F0) {
        rc = flush(j);
        if (rc == 0) rc = lpgp_gram_assemble_grid(ctx, e.kd, e.ngroups, e.F0, j == bi ? nullptr : e.F1, mat, bi, j);
        continue;
      }
      const lpgp_pts* Xc = (j == bi) ? X_new : e.X1;
      const lpgp_block& Bj = mat->blocks[j];
      // (no early return in here: a failure must reach the roll-back of the new block below)
      if (!(X_new && Xc && e.kd)) {
        set_error("lpgp_mat_condition: null point set or descriptor in row entry %d", j);
        rc = -1;
        break;
      }
      if (!(X_new->n == Bi.n && Xc->n == Bj.n && Xc->d == X_new->d && e.kd[0].d == X_new->d)) {
        set_error("lpgp_mat_condition: shape mismatch in row entry %d", j);
        rc = -1;
        break;
      }
      rc = lower_kdesc(e.kd, e.ngroups, &descs[(size_t)j]);
      if (rc != 0) break;
      if (run_start >= 0 && !assemble_same_fast(descs[(size_t)run_start], descs[(size_t)j])) rc = flush(j);
      if (rc != 0) break;
      if (run_start < 0) run_start = j;
      jobs.push_back(AsmJob{X_new->x, X_new->n, X_new->n_pad, Xc->x, Xc->n, Xc->n_pad, Bi.poff, Bj.poff, j == bi ? 1 : 0});
    }
    if (rc == 0) rc = flush(nrow);
  } else
  for (int j = 0; j < nrow && rc == 0; ++j) {
    const lpgp_cond_block& e = row[j];
    if (e.F0)
      rc = lpgp_gram_assemble_grid(ctx, e.kd, e.ngroups, e.F0, j == bi ? nullptr : e.F1, mat, bi, j);
    else
      rc = lpgp_gram_assemble(ctx, e.kd, e.ngroups, X_new, j == bi ? nullptr : e.X1, mat, bi, j);
  }
  if (rc == 0 && noise_scalar != 0.0 && !fused_finish) rc = lpgp_mat_add_diag(ctx, mat, bi, nullptr, noise_scalar);
  if (rc == 0 && noise_dense) rc = lpgp_mat_add_dense(ctx, mat, bi, noise_dense);
  if (rc == 0 && fused_finish && !noise_diag) {
    const lpgp_block& B = mat->blocks[bi];
    launch_finish_block(ctx->s_main, mat->a, mat->lr_cap, B.poff + B.n, B.pn - B.n, mat->cap, mat_layout(ctx), B.poff, noise_scalar != 0.0 ? B.n : 0,
                        nullptr, noise_scalar);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      set_error("lpgp_mat_condition: %s", hipGetErrorString(e));
      rc = -1;
    }
  }
  if (rc == 0 && noise_diag) {
    // the vector is staged into the block's own segment of the weights buffer (unused until the first solve, which is
    // stream-ordered behind the kernel below) through a stream that is idle during conditionings: nothing waits for the
    // panel stream, where an enqueued factorisation of the previous block may still be running
    const lpgp_block& B = mat->blocks[bi];
    // (only while an enqueued factorisation is in flight -- `unchecked` -- is the side stream worth anything; otherwise the
    //  panel stream, which orders the upload behind whatever a caller of the C API left running on it: lpgp.h, "streams")
    hipStream_t sc = (mat->unchecked && ctx->s_upd_all && !ctx->single_stream && !ctx->distributed()) ? ctx->s_upd_all : ctx->s_main;
    hipError_t e = hipMemcpyAsync(mat->w + B.poff, noise_diag, (size_t)B.n * sizeof(double), hipMemcpyHostToDevice, sc);
    if (e == hipSuccess) e = hipStreamSynchronize(sc);
    if (e != hipSuccess) {
      set_error("lpgp_mat_condition: noise upload: %s", hipGetErrorString(e));
      rc = -1;
    } else if (fused_finish) {
      launch_finish_block(ctx->s_main, mat->a, mat->lr_cap, B.poff + B.n, B.pn - B.n, mat->cap, mat_layout(ctx), B.poff, B.n, mat->w + B.poff, 0.0);
      if ((e = hipGetLastError()) != hipSuccess) {
        set_error("lpgp_mat_condition: %s", hipGetErrorString(e));
        rc = -1;
      }
    } else {
      rc = launch_add_diag(ctx->s_main, mat->a, mat->lr_cap, B.poff, B.n, mat->w + B.poff, 0.0, mat_layout(ctx));
    }
  }
  int32_t h = 0;
  // lazy == 2: the block row is assembled and the factorisation is left to whoever needs the factor first -- the next
  // lpgp_mat_condition / lpgp_potrf / lpgp_potrf_enqueue, or lpgp_potrf_predict, which lets the prediction ride inside it
  if (rc == 0 && lazy != 2) rc = lazy ? lpgp_potrf_enqueue(ctx, mat) : lpgp_potrf(ctx, mat, &h);
  if (rc != 0 || h != 0) {
    // keep the error text of the failing step: lpgp_mat_pop_block succeeds and would not touch it, but be explicit
    const std::string msg = lpgp::last_error();
    if (lpgp_mat_pop_block(ctx, mat) != 0 || rc != 0) set_error("%s", msg.c_str());
  }
  if (info) *info = h;
  return rc;
}

int lpgp_mat_check(lpgp_ctx* ctx, lpgp_mat* mat, int32_t* info, int32_t* block) {
  LPGP_CHECK(ctx && mat && info, "lpgp_mat_check: null argument");
  LPGP_DEVICE(ctx);
  *info = 0;
  if (block) *block = -1;
  if (!mat->unchecked) return 0;
  int h = 0;
  if (mat->status_known) {
    h = mat->status_value;             // read back by lpgp_potrf_predict together with its results, nothing enqueued since
  } else {
    LPGP_HIP(hipMemcpyAsync(&h, mat->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->s_main));
    LPGP_HIP(hipStreamSynchronize(ctx->s_main));
  }
  mat->status_known = 0;
  mat->unchecked = 0;
  if (h < 0) mat->poisoned = 1;      // (not a pivot: the panel's contents are undefined and the status word stays at INT_MIN, so no later pivot failure could be recorded either)
  LPGP_CHECK(h >= 0, "resident panel chain: a hand-over between workgroups timed out (device status %d); set LPGP_CHAIN_RESIDENT=-1", h);
  *info = h;
  if (h > 0 && block) {
    std::vector<lpgp_block> all = mat->blocks;
    all.insert(all.end(), mat->hidden.begin(), mat->hidden.end());
    for (size_t b = 0; b < all.size(); ++b)
      if ((int64_t)h - 1 >= all[b].poff && (int64_t)h - 1 < all[b].poff + all[b].pn) *block = (int32_t)b;
  }
  return 0;
}

int lpgp_mat_truncate(lpgp_ctx* ctx, lpgp_mat* mat, int32_t nblocks) {
  LPGP_CHECK(ctx && mat, "lpgp_mat_truncate: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(mat->hidden.empty(), "lpgp_mat_truncate: a strict prefix of the blocks is in view");
  LPGP_CHECK(nblocks >= 0 && nblocks <= (int32_t)mat->blocks.size(), "lpgp_mat_truncate: %d of %d blocks", nblocks, (int)mat->blocks.size());
  LPGP_CHECK(!ctx->distributed(), "lpgp_mat_truncate: single GPU only");
  LPGP_HIP(hipStreamSynchronize(ctx->s_main));        // nothing of the dropped blocks still in flight
  mat->blocks.resize((size_t)nblocks);
  mat->n = nblocks ? mat->blocks.back().off + mat->blocks.back().n : 0;
  mat->pn = nblocks ? mat->blocks.back().poff + mat->blocks.back().pn : 0;
  if (mat->pn_fact > mat->pn) mat->pn_fact = mat->pn;
  LPGP_HIP(hipMemset(mat->d_status, 0, sizeof(int)));
  mat->unchecked = 0;
  mat->status_known = 0;
  mat->has_w = 0;
  mat->has_r = 0;
  return 0;
}

int lpgp_potrs(lpgp_ctx* ctx, lpgp_mat* mat, double* b_host, int64_t nrhs) {
  LPGP_CHECK(ctx && mat && b_host && nrhs >= 1, "lpgp_potrs: bad argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_potrs");
  LPGP_CHECK(mat->pn_fact == mat->pn && mat->pn > 0, "lpgp_potrs: matrix is not factored");
  const int64_t pn = mat->pn, n = mat->n, m_pad = round_up(nrhs, TILE);
  double* dv = nullptr;
  LPGP_HIP(hipMalloc(&dv, (size_t)pn * m_pad * sizeof(double)));
  std::vector<double> hp((size_t)pn * m_pad, 0.0);
  for (int64_t j = 0; j < nrhs; ++j) scatter_padded(mat, b_host + j * n, hp.data() + j * pn);
  int rc = 0;
  do {
    if (hipMemcpyAsync(dv, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice, ctx->s_main) != hipSuccess) { rc = -1; break; }
    rc = ctx->distributed() ? dist_fail(ctx, trsm_lower_dist(ctx, mat, pn / TILE, dv, pn, m_pad)) : trsm_lower_blocked(ctx, mat, pn / TILE, dv, pn, m_pad);
    if (rc) break;
    rc = ctx->distributed() ? dist_fail(ctx, trsm_lower_t_dist(ctx, mat, pn / TILE, dv, pn, m_pad)) : trsm_lower_t_blocked(ctx, mat, pn / TILE, dv, pn, m_pad);
    if (rc) break;
    // (wait FIRST, watched: the copy into pageable memory below is synchronous and would wait unwatched behind a streamed
    //  solve whose peer is gone)
    if ((rc = sync_stream(ctx, ctx->s_main)) != 0) break;
    if (hipMemcpyAsync(hp.data(), dv, hp.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess) { rc = -1; break; }
    rc = sync_stream(ctx, ctx->s_main);          // (behind a streamed solve: polls, so that a failed peer cannot block this rank)
  } while (0);
  (void)hipFree(dv);
  if (rc != 0) {
    if (rc == -1) set_error("lpgp_potrs: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
    return rc;
  }
  for (int64_t j = 0; j < nrhs; ++j) gather_padded(mat, hp.data() + j * pn, b_host + j * n);
  return 0;
}

int lpgp_solve_weights(lpgp_ctx* ctx, lpgp_mat* mat, const double* r_host, double* w_host) {
  LPGP_CHECK(ctx && mat && r_host, "lpgp_solve_weights: null argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_solve_weights");
  LPGP_CHECK(mat->pn_fact == mat->pn && mat->pn > 0, "lpgp_solve_weights: matrix is not factored");
  const int64_t pn = mat->pn;
  int rc = ensure_tmp(ctx, pn + 2);                 // (+ 2: the ticket words of the resident solve, trsv.hip)
  if (rc != 0) return rc;
  // through pinned staging: the residual up and the weights + the solve's status word back are asynchronous copies, ONE wait
  // (round 6; until then two blocking copies through pageable memory around 2 x T dependent launches)
  rc = ensure_stage(ctx, pn);
  if (rc != 0) return rc;
  double* const hs = ctx->h_stage;
  int* const hinfo = ctx->h_info_pinned + 6;
  scatter_padded(mat, r_host, hs);
  LPGP_HIP(hipMemcpyAsync(mat->w, hs, (size_t)pn * sizeof(double), hipMemcpyHostToDevice, ctx->s_main));
  if (ctx->distributed()) {
    // multi-GPU: the factor is streamed; the vector rides as column 0 of a 128-column block on every rank (all
    // ranks end with the same weights, no further communication)
    void* pv = nullptr;
    const size_t vb = (size_t)pn * TILE * sizeof(double);
    if (pool_alloc(ctx, &pv, vb, nullptr) != 0) return dist_fail(ctx, -1);      // (the peers are about to enter the streamed solve)
    double* dv = (double*)pv;
    hipError_t e = hipMemsetAsync(dv, 0, vb, ctx->s_main);
    if (e == hipSuccess) e = hipMemcpyAsync(dv, mat->w, (size_t)pn * sizeof(double), hipMemcpyDeviceToDevice, ctx->s_main);
    rc = dist_fail(ctx, e == hipSuccess ? trsm_lower_dist(ctx, mat, pn / TILE, dv, pn, TILE) : -1);
    if (rc == 0) rc = dist_fail(ctx, trsm_lower_t_dist(ctx, mat, pn / TILE, dv, pn, TILE));
    if (rc == 0 && hipMemcpyAsync(mat->w, dv, (size_t)pn * sizeof(double), hipMemcpyDeviceToDevice, ctx->s_main) != hipSuccess) rc = dist_fail(ctx, -1);
    pool_free(ctx, pv, vb);
    *hinfo = 0;
  } else {
    LPGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->s_main));
    rc = solve_vec(ctx, mat, pn / TILE, mat->w, ctx->d_tmp, ctx->d_info);
    if (rc == 0) LPGP_HIP(hipMemcpyAsync(hinfo, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->s_main));
  }
  if (rc != 0) return rc;
  if (ctx->distributed()) LPGP_TRY(sync_stream(ctx, ctx->s_main));      // (watched: a copy behind a streamed solve whose peer is gone would wait for ever)
  LPGP_HIP(hipMemcpyAsync(hs, mat->w, (size_t)pn * sizeof(double), hipMemcpyDeviceToHost, ctx->s_main));
  LPGP_TRY(sync_stream(ctx, ctx->s_main));
  LPGP_CHECK(*hinfo >= 0, "resident single-vector solve: a hand-over between workgroups timed out (device status %d); set LPGP_TRSV_RESIDENT=0", *hinfo);
  mat->has_w = 1;
  if (w_host) gather_padded(mat, hs, w_host);
  return 0;
}

int lpgp_mat_set_residual(lpgp_ctx* ctx, lpgp_mat* mat, const double* r_host) {
  LPGP_CHECK(ctx && mat && r_host, "lpgp_mat_set_residual: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(mat->pn > 0, "lpgp_mat_set_residual: empty matrix");      // (factored or not: lpgp_potrf_predict takes it along)
  if (!ctx->distributed() && !ctx->single_stream) {
    // Through pinned staging on the panel stream, NO wait (round 5): every reader of the residual is a kernel or a copy on the
    // panel stream (lpgp_predict, lpgp_potrf_predict, lpgp_solve_weights), the caller's vector has been consumed when this
    // returns; the staging buffer is reused only after the previous upload has completed (an event, long signalled when the
    // next call comes).  Until round 5: a copy from pageable memory on a side stream and a wait for it, ~15 us of a host that a
    // small problem's step is bound by.
    if (ctx->stage_r_cap < mat->pn) {
      if (ctx->h_stage_r) {
        LPGP_HIP(hipEventSynchronize(ctx->ev_stage_r));
        LPGP_HIP(hipHostFree(ctx->h_stage_r));
      }
      ctx->h_stage_r = nullptr;
      ctx->stage_r_cap = 0;
      LPGP_HIP(hipHostMalloc(&ctx->h_stage_r, (size_t)mat->pn * sizeof(double), hipHostMallocDefault));
      ctx->stage_r_cap = mat->pn;
      if (!ctx->ev_stage_r) LPGP_HIP(hipEventCreateWithFlags(&ctx->ev_stage_r, hipEventDisableTiming));
    } else {
      LPGP_HIP(hipEventSynchronize(ctx->ev_stage_r));
    }
    scatter_padded(mat, r_host, ctx->h_stage_r);
    LPGP_HIP(hipMemcpyAsync(mat->r(), ctx->h_stage_r, (size_t)mat->pn * sizeof(double), hipMemcpyHostToDevice, ctx->s_main));
    LPGP_HIP(hipEventRecord(ctx->ev_stage_r, ctx->s_main));
    mat->has_r = 1;
    return 0;
  }
  LPGP_CHECK(mat->pn_fact == mat->pn, "lpgp_mat_set_residual: matrix is not factored");      // (this branch has no lpgp_potrf_predict to consume an unfactored one)
  std::vector<double> hp((size_t)mat->pn);
  scatter_padded(mat, r_host, hp.data());
  // NOT on the panel stream: the residual's place in HBM is touched by no kernel of the factorisation, so the upload need
  // not queue behind an enqueued factorisation (lpgp_potrf_enqueue); it is complete when this call returns.  The update
  // stream of the blocked solves is idle whenever this is called (its work is joined into the panel stream at the end of
  // every solve) -- and it is an EXISTING stream: HIP multiplexes streams over four hardware queues, and a fifth stream
  // made two of them share one (measured: c2 predict 2.8 -> 5.6 ms).
  hipStream_t sc = (mat->unchecked && ctx->s_upd_all && !ctx->single_stream) ? ctx->s_upd_all : ctx->s_main;
  LPGP_HIP(hipMemcpyAsync(mat->r(), hp.data(), (size_t)mat->pn * sizeof(double), hipMemcpyHostToDevice, sc));
  LPGP_HIP(hipStreamSynchronize(sc));
  mat->has_r = 1;
  return 0;
}

// ---- prediction -----------------------------------------------------------------------------
// What lpgp_rhs_create has to clear -- the rows of the blocks' padding tails (all logical columns) and the spare columns (all
// rows) -- in ONE launch (until round 5: a memset for the spare columns and a 2-D memset per padded block; five launches in
// front of every prediction of the reference's 2-D Poisson problem).  Job j < nj: rows [row0[j], row0[j] + nr[j]) x m columns;
// job nj: ld x (m_pad - m) doubles from column m on.
constexpr int RHS_PAD_JOBS = 15;
struct RhsPadJobs { int64_t row0[RHS_PAD_JOBS]; int32_t nr[RHS_PAD_JOBS]; int32_t nj; };
__global__ void rhs_pad_kernel(double* v, int64_t ld, int64_t m, int64_t m_pad, RhsPadJobs jb, int spare) {
  const int j = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j < jb.nj) {
    const int64_t nr = jb.nr[j], tot = nr * m;
    double* p = v + jb.row0[j];
    for (int64_t t = t0; t < tot; t += stride) p[(t % nr) + (t / nr) * ld] = 0.0;
  } else if (spare) {
    const int64_t tot = ld * (m_pad - m);
    double* p = v + ld * m;
    for (int64_t t = t0; t < tot; t += stride) p[t] = 0.0;
  }
}

int lpgp_rhs_create(lpgp_ctx* ctx, const lpgp_mat* mat, int64_t m, lpgp_rhs** out) {
  LPGP_CHECK(ctx && mat && out && m >= 1, "lpgp_rhs_create: bad argument");
  LPGP_DEVICE(ctx);
  lpgp_rhs* r = new lpgp_rhs();
  r->ctx = ctx;
  r->ld = mat->pn;
  r->m = m;
  r->m_pad = round_up(m + 1, TILE);       // at least one spare column (see lpgp_predict)
  r->v = nullptr;
  void* pv = nullptr;
  if (pool_alloc(ctx, &pv, (size_t)r->ld * r->m_pad * sizeof(double), nullptr) != 0) {
    delete r;
    return -1;
  }
  r->v = (double*)pv;
  // lpgp_cross_assemble writes every logical row of every logical column: only the spare columns and
  // the rows of the blocks' padding tails have to be cleared (c3: 17 MB instead of 571 MB per prediction)
  {
    RhsPadJobs jb;
    jb.nj = 0;
    int spare = 1;
    int64_t work = r->ld * (r->m_pad - m);
    auto flush = [&]() {
      const int64_t wgs = (work + 2047) / 2048;
      hipLaunchKernelGGL(rhs_pad_kernel, dim3((unsigned)(wgs < 1 ? 1 : (wgs > 2048 ? 2048 : wgs)), (unsigned)(jb.nj + spare)), dim3(256), 0, ctx->s_main,
                         r->v, r->ld, m, r->m_pad, jb, spare);
      jb.nj = 0;
      spare = 0;
      work = 0;
    };
    for (const auto& b : mat->blocks)
      if (b.pn > b.n) {
        jb.row0[jb.nj] = b.poff + b.n;
        jb.nr[jb.nj] = (int32_t)(b.pn - b.n);
        ++jb.nj;
        work = std::max(work, (int64_t)(b.pn - b.n) * m);
        if (jb.nj == RHS_PAD_JOBS) flush();
      }
    if (jb.nj > 0 || spare) flush();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      set_error("lpgp_rhs_create: %s", hipGetErrorString(e));
      pool_free(ctx, r->v, (size_t)r->ld * r->m_pad * sizeof(double));
      delete r;
      return -1;
    }
  }
  r->assembled.assign(mat->blocks.size(), 0);
  ++ctx->live_mats;
  *out = r;
  return 0;
}

// rows of observation blocks that were never assembled count as zero cross-covariance
static int rhs_clear_unassembled(lpgp_ctx* ctx, const lpgp_mat* mat, lpgp_rhs* r) {
  for (size_t bi = 0; bi < r->assembled.size() && bi < mat->blocks.size(); ++bi) {
    if (r->assembled[bi]) continue;
    const lpgp_block& b = mat->blocks[bi];
    if (b.n > 0)
      LPGP_HIP(hipMemset2DAsync(r->v + b.poff, (size_t)r->ld * sizeof(double), 0, (size_t)b.n * sizeof(double), (size_t)r->m,
                                ctx->s_main));
    r->assembled[bi] = 1;
  }
  return 0;
}

int lpgp_rhs_destroy(lpgp_rhs* r) {
  if (!r) return 0;
  (void)hipSetDevice(r->ctx->device);
  pool_free(r->ctx, r->v, (size_t)r->ld * r->m_pad * sizeof(double));
  --r->ctx->live_mats;
  delete r;
  return 0;
}

int lpgp_cross_assemble(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups, const lpgp_pts* X_obs,
                        const lpgp_pts* X_test, lpgp_rhs* rhs, const lpgp_mat* mat, int32_t bi) {
  LPGP_CHECK(ctx && kd && X_obs && X_test && rhs && mat, "lpgp_cross_assemble: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(bi >= 0 && bi < (int)mat->blocks.size(), "lpgp_cross_assemble: bad block %d", bi);
  const lpgp_block& B = mat->blocks[bi];
  LPGP_CHECK(X_obs->n == B.n && X_test->n == rhs->m && X_obs->d == X_test->d && kd[0].d == X_obs->d,
             "lpgp_cross_assemble: shape mismatch");
  LPGP_CHECK(rhs->ld == mat->pn, "lpgp_cross_assemble: rhs was created for a different matrix size");
  DevDesc desc;
  int rc = lower_kdesc(kd, ngroups, &desc);
  if (rc != 0) return rc;
  rc = launch_assemble(ctx, ctx->s_main, desc, X_obs->x, X_obs->n, X_obs->n_pad, X_test->x, X_test->n,
                       X_test->n_pad, rhs->v, rhs->ld, B.poff, 0, 0);
  if (rc == 0 && bi < (int)rhs->assembled.size()) rhs->assembled[bi] = 1;
  return rc;       // asynchronous (see lpgp_gram_assemble)
}

int lpgp_cross_assemble_row(lpgp_ctx* ctx, const lpgp_cross_block* blocks, int32_t nblocks, const lpgp_pts* X_test, lpgp_rhs* rhs,
                            const lpgp_mat* mat) {
  LPGP_CHECK(ctx && blocks && X_test && rhs && mat, "lpgp_cross_assemble_row: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(nblocks == (int32_t)mat->blocks.size(), "lpgp_cross_assemble_row: %d entries for %d blocks", nblocks, (int)mat->blocks.size());
  LPGP_CHECK(rhs->ld == mat->pn && X_test->n == rhs->m, "lpgp_cross_assemble_row: rhs was created for a different matrix size or point count");
  std::vector<DevDesc> descs((size_t)nblocks);
  std::vector<AsmJob> jobs;
  int run_start = -1, rc = 0;
  auto flush = [&]() -> int {
    int r = 0;
    if (run_start >= 0 && !jobs.empty())
      r = launch_assemble_batch(ctx, ctx->s_main, descs[(size_t)run_start], jobs.data(), (int)jobs.size(), rhs->v, rhs->ld, Layout2D());
    run_start = -1;
    jobs.clear();
    return r;
  };
  for (int bi = 0; bi < nblocks && rc == 0; ++bi) {
    const lpgp_cross_block& e = blocks[bi];
    const lpgp_block& B = mat->blocks[(size_t)bi];
    if (!e.kd || !e.X_obs) continue;                       // a block without cross-covariance: its rows stay zero (rhs_clear_unassembled)
    LPGP_CHECK(e.X_obs->n == B.n && e.X_obs->d == X_test->d && e.kd[0].d == X_test->d, "lpgp_cross_assemble_row: shape mismatch in block %d", bi);
    rc = lower_kdesc(e.kd, e.ngroups, &descs[(size_t)bi]);
    if (rc != 0) break;
    if (run_start >= 0 && !assemble_same_fast(descs[(size_t)run_start], descs[(size_t)bi])) rc = flush();
    if (rc != 0) break;
    if (run_start < 0) run_start = bi;
    jobs.push_back(AsmJob{e.X_obs->x, e.X_obs->n, e.X_obs->n_pad, X_test->x, X_test->n, X_test->n_pad, B.poff, 0, 0});
    if (bi < (int)rhs->assembled.size()) rhs->assembled[(size_t)bi] = 1;
  }
  if (rc == 0) rc = flush();
  return rc;       // asynchronous (see lpgp_gram_assemble)
}

int lpgp_trsm_lower(lpgp_ctx* ctx, lpgp_mat* mat, lpgp_rhs* V) {
  LPGP_CHECK(ctx && mat && V, "lpgp_trsm_lower: null argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_trsm_lower");
  LPGP_CHECK(mat->pn_fact == mat->pn && V->ld == mat->pn, "lpgp_trsm_lower: matrix not factored or size mismatch");
  int rc = rhs_clear_unassembled(ctx, mat, V);
  if (rc != 0) return rc;
  rc = ctx->distributed() ? dist_fail(ctx, trsm_lower_dist(ctx, mat, mat->pn / TILE, V->v, V->ld, V->m_pad))
                          : trsm_lower_blocked(ctx, mat, mat->pn / TILE, V->v, V->ld, V->m_pad);
  if (rc != 0) return rc;
  LPGP_TRY(sync_stream(ctx, ctx->s_main));
  return 0;
}

int lpgp_predict(lpgp_ctx* ctx, lpgp_mat* mat, lpgp_rhs* K, const double* prior_mean_host,
                 const double* kxx_host, double* mean_host, double* var_host) {
  LPGP_CHECK(ctx && mat && K, "lpgp_predict: null argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_predict");
  LPGP_CHECK(mat->pn_fact == mat->pn && K->ld == mat->pn, "lpgp_predict: matrix not factored or size mismatch");
  const int64_t m = K->m;
  int rc = ensure_tmp(ctx, 2 * K->m_pad);
  if (rc != 0) return rc;
  rc = rhs_clear_unassembled(ctx, mat, K);
  if (rc != 0) return rc;
  // results come back through pinned staging: the copy is asynchronous (the one wait polls, so that a failed peer of a
  // multi-GPU job cannot block this rank inside a copy) and mean + variance are ONE copy, not two blocking round trips
  rc = ensure_stage(ctx, 2 * K->m_pad);
  if (rc != 0) return rc;
  double* const hs = ctx->h_stage;
  // Mean and variance together: the variance needs V = L^{-1} K_Xx anyway, and
  //   K_xX G^{-1} r = V^T z  with  z = L^{-1} r,
  // so no representer weights are needed: the residual rides through the blocked solve as
  // the spare column m of the right-hand side (a side-stream triangular solve for z was
  // measured instead: its 132 dependent launches queue behind the resident GEMM workgroups
  // and stretch the step from 68 to 100 ms).
  const bool via_z = mean_host && var_host && !mat->has_w && mat->has_r;
  if (mean_host && !via_z) {
    LPGP_CHECK(mat->has_w, "lpgp_predict: representer weights not computed (call lpgp_solve_weights%s)",
               var_host ? " or lpgp_mat_set_residual" : "");
    hipLaunchKernelGGL(col_reduce_kernel, dim3((unsigned)m), dim3(256), 0, ctx->s_main, K->v, K->ld, mat->pn,
                       (const double*)mat->w, ctx->d_tmp);
    LPGP_HIP(hipGetLastError());
    LPGP_HIP(hipMemcpyAsync(hs, ctx->d_tmp, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, ctx->s_main));
    LPGP_TRY(sync_stream(ctx, ctx->s_main));
    for (int64_t j = 0; j < m; ++j) mean_host[j] = (prior_mean_host ? prior_mean_host[j] : 0.0) + hs[j];
  }
  if (var_host) {
    LPGP_CHECK(kxx_host != nullptr, "lpgp_predict: kxx_host required for the variance");
    double* zcol = K->v + (int64_t)m * K->ld;
    if (via_z)
      LPGP_HIP(hipMemcpyAsync(zcol, mat->r(), (size_t)mat->pn * sizeof(double), hipMemcpyDeviceToDevice, ctx->s_main));
    rc = ctx->distributed() ? dist_fail(ctx, trsm_lower_dist(ctx, mat, mat->pn / TILE, K->v, K->ld, K->m_pad))
                            : trsm_lower_blocked(ctx, mat, mat->pn / TILE, K->v, K->ld, K->m_pad);
    if (rc != 0) return rc;
    if (via_z) {
      hipLaunchKernelGGL(col_reduce2_kernel, dim3((unsigned)m), dim3(256), 0, ctx->s_main, K->v, K->ld, mat->pn,
                         (const double*)zcol, ctx->d_tmp, ctx->d_tmp + K->m_pad);
      LPGP_HIP(hipGetLastError());
      LPGP_HIP(hipMemcpyAsync(hs, ctx->d_tmp, (size_t)(K->m_pad + m) * sizeof(double), hipMemcpyDeviceToHost, ctx->s_main));
      LPGP_TRY(sync_stream(ctx, ctx->s_main));
      for (int64_t j = 0; j < m; ++j) mean_host[j] = (prior_mean_host ? prior_mean_host[j] : 0.0) + hs[j];
      for (int64_t j = 0; j < m; ++j) var_host[j] = kxx_host[j] - hs[K->m_pad + j];
    } else {
      hipLaunchKernelGGL(col_reduce_kernel, dim3((unsigned)m), dim3(256), 0, ctx->s_main, K->v, K->ld, mat->pn,
                         (const double*)nullptr, ctx->d_tmp);
      LPGP_HIP(hipGetLastError());
      LPGP_HIP(hipMemcpyAsync(hs, ctx->d_tmp, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, ctx->s_main));
      LPGP_TRY(sync_stream(ctx, ctx->s_main));
      for (int64_t j = 0; j < m; ++j) var_host[j] = kxx_host[j] - hs[j];
    }
  }
  return 0;
}

int lpgp_potrf_predict(lpgp_ctx* ctx, lpgp_mat* mat, lpgp_rhs* K, const double* prior_mean_host, const double* kxx_host,
                       double* mean_host, double* var_host) {
  LPGP_CHECK(ctx && mat && K && kxx_host && mean_host && var_host, "lpgp_potrf_predict: null argument");
  LPGP_DEVICE(ctx);
  LPGP_MAT_ALIVE(mat, "lpgp_potrf_predict");
  LPGP_CHECK(!ctx->distributed(), "lpgp_potrf_predict: single GPU only (a multi-GPU job factors collectively: lpgp_potrf, lpgp_predict)");
  LPGP_CHECK(mat->hidden.empty(), "lpgp_potrf_predict: a strict prefix of the blocks is in view");
  LPGP_CHECK(K->ld == mat->pn && mat->pn > 0, "lpgp_potrf_predict: right-hand side built for another matrix size");
  LPGP_CHECK(mat->has_r, "lpgp_potrf_predict: no residual resident (lpgp_mat_set_residual)");
  const int64_t m = K->m;
  int rc = ensure_tmp(ctx, 2 * K->m_pad);
  if (rc != 0) return rc;
  rc = ensure_stage(ctx, 2 * K->m_pad);
  if (rc != 0) return rc;
  rc = rhs_clear_unassembled(ctx, mat, K);
  if (rc != 0) return rc;
  double* zcol = K->v + (int64_t)m * K->ld;             // the residual rides as the spare column: z = L^{-1} r, mean = V^T z
  LPGP_HIP(hipMemcpyAsync(zcol, mat->r(), (size_t)mat->pn * sizeof(double), hipMemcpyDeviceToDevice, ctx->s_main));
  const bool fresh = mat->pn_fact < mat->pn;
  ctx->d_info_cur = mat->d_status;
  rc = potrf_predict_blocked(ctx, mat, mat->pn_fact / TILE, mat->pn / TILE, K->v, K->ld, K->m_pad);
  if (rc != 0) {
    if (fresh) mat->poisoned = 1;    // (partly factored in place with pn_fact unchanged -- a retry would factor a half-factored matrix)
    return rc;
  }
  if (fresh) {
    mat->pn_fact = mat->pn;          // provisionally, as lpgp_potrf_enqueue: lpgp_mat_check / lpgp_mat_truncate take it back on failure
    mat->unchecked = 1;
    mat->has_w = 0;
  }
  mat->status_known = 0;
  hipLaunchKernelGGL(col_reduce2_kernel, dim3((unsigned)m), dim3(256), 0, ctx->s_main, K->v, K->ld, mat->pn, (const double*)zcol, ctx->d_tmp,
                     ctx->d_tmp + K->m_pad);
  LPGP_HIP(hipGetLastError());
  // mean, variance and the status word of the factorisation come back together: two asynchronous copies into pinned memory,
  // ONE wait (until round 5: two blocking copies into pageable memory here and a third in lpgp_mat_check, ~25 us each --
  // a tenth of a step of the reference's own problem sizes)
  double* const hs = ctx->h_stage;
  int* const hinfo = ctx->h_info_pinned + 4;
  LPGP_HIP(hipMemcpyAsync(hs, ctx->d_tmp, (size_t)(K->m_pad + m) * sizeof(double), hipMemcpyDeviceToHost, ctx->s_main));
  LPGP_HIP(hipMemcpyAsync(hinfo, mat->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->s_main));
  LPGP_TRY(sync_stream(ctx, ctx->s_main));
  if (mat->unchecked) {
    mat->status_value = *hinfo;
    mat->status_known = 1;
  }
  for (int64_t j = 0; j < m; ++j) {
    mean_host[j] = (prior_mean_host ? prior_mean_host[j] : 0.0) + hs[j];
    var_host[j] = kxx_host[j] - hs[K->m_pad + j];
  }
  return 0;
}

int lpgp_rhs_inner(lpgp_ctx* ctx, lpgp_rhs* A, lpgp_rhs* B, double* out_host) {
  LPGP_CHECK(ctx && A && B && out_host, "lpgp_rhs_inner: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(A->ld == B->ld, "lpgp_rhs_inner: row mismatch");
  const int64_t ma = A->m_pad, mb = B->m_pad;
  double* dc = nullptr;
  LPGP_HIP(hipMalloc(&dc, (size_t)ma * mb * sizeof(double)));
  GemmArgs g;
  g.A = A->v; g.B = B->v; g.C = dc; g.lda = A->ld; g.ldb = B->ld; g.ldc = ma;
  g.mt = (int)(ma / TILE); g.nt = (int)(mb / TILE); g.k = (int)A->ld; g.alpha = 1.0; g.beta = 0.0;
  g.tri = 0;
  int rc = launch_gemm(ctx, ctx->s_main, 1, 1, g, LPGP_K_GEMM);
  std::vector<double> h((size_t)ma * mb);
  if (rc == 0 && hipMemcpyAsync(h.data(), dc, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess) rc = -1;
  if (rc == 0 && hipStreamSynchronize(ctx->s_main) != hipSuccess) rc = -1;
  (void)hipFree(dc);
  if (rc != 0) return rc;
  for (int64_t i = 0; i < A->m; ++i)
    for (int64_t j = 0; j < B->m; ++j) out_host[i * B->m + j] = h[(size_t)(i + j * ma)];
  return 0;
}

int lpgp_rhs_matmul(lpgp_ctx* ctx, const lpgp_rhs* A, const double* B_host, int64_t m, lpgp_rhs** out_new) {
  LPGP_CHECK(ctx && A && B_host && out_new && m >= 1, "lpgp_rhs_matmul: bad argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(!ctx->distributed(), "lpgp_rhs_matmul: single GPU only");
  // the result shares A's row layout whatever the matrix A was created for has become since (a posterior is a value: later
  // conditionings extend the shared matrix); every element of it is written by the product (beta = 0): no clearing
  lpgp_rhs* out = new lpgp_rhs();
  out->ctx = ctx;
  out->ld = A->ld;
  out->m = m;
  out->m_pad = round_up(m + 1, TILE);
  out->v = nullptr;
  {
    void* pv = nullptr;
    if (pool_alloc(ctx, &pv, (size_t)out->ld * out->m_pad * sizeof(double), nullptr) != 0) {
      delete out;
      return -1;
    }
    out->v = (double*)pv;
  }
  out->assembled.assign(A->assembled.size(), 1);
  ++ctx->live_mats;
  const int64_t ka = A->m_pad, mp = out->m_pad;
  // B, zero-padded to (columns of A incl. its spare ones) x (columns of out incl. its spare ones), column-major: the spare
  // columns of A meet zero rows, the spare columns of out come out as zeros
  std::vector<double> hb((size_t)ka * mp, 0.0);
  for (int64_t kk = 0; kk < A->m; ++kk)
    for (int64_t j = 0; j < m; ++j) hb[(size_t)(kk + j * ka)] = B_host[kk * m + j];
  void* pb = nullptr;
  const size_t bb = hb.size() * sizeof(double);
  if (pool_alloc(ctx, &pb, bb, nullptr) != 0) {
    (void)lpgp_rhs_destroy(out);
    return -1;
  }
  int rc = 0;
  if (hipMemcpyAsync(pb, hb.data(), bb, hipMemcpyHostToDevice, ctx->s_main) != hipSuccess) rc = -1;
  if (rc == 0) {
    GemmArgs g;
    g.A = A->v; g.B = (const double*)pb; g.C = out->v; g.lda = A->ld; g.ldb = ka; g.ldc = out->ld;
    g.mt = (int)(out->ld / TILE); g.nt = (int)(mp / TILE); g.k = (int)ka; g.alpha = 1.0; g.beta = 0.0;
    g.tri = 0;
    rc = launch_gemm(ctx, ctx->s_main, 0, 1, g, LPGP_K_GEMM);
  }
  if (rc == 0 && hipStreamSynchronize(ctx->s_main) != hipSuccess) rc = -1;      // (the staging vector is borrowed by the copy)
  pool_free(ctx, pb, bb);
  if (rc != 0) {
    if (rc == -1) set_error("lpgp_rhs_matmul: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
    (void)lpgp_rhs_destroy(out);
    return rc;
  }
  *out_new = out;
  return 0;
}

int lpgp_gemm_host(lpgp_ctx* ctx, int32_t transa, int32_t transb, int64_t m, int64_t n, int64_t k, double alpha,
                   const double* A_host, const double* B_host, double beta, double* C_host) {
  LPGP_CHECK(ctx && A_host && B_host && C_host && m >= 1 && n >= 1 && k >= 1, "lpgp_gemm_host: bad argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(!ctx->distributed(), "lpgp_gemm_host: single GPU only");
  // The kernel computes column-major  C' = op(A') op(B')  on whole 128 x 128 tiles.  A C-order m x n result is the column-major
  // n x m matrix C^T = op(B)^T op(A)^T in the same memory: A' = op(B)^T (n x k), B' = op(A)^T (k x m), both read in place --
  // a C-order k x n array IS column-major n x k (ta = 0), a C-order n x k one is its k-fastest form (ta = 1); likewise for B'.
  const int64_t np_ = round_up(n, TILE), mp = round_up(m, TILE), kp = round_up(k, 16);
  // zero-padded device images (rows beyond the logical extent must be zeros, not stale pool contents: 0 * NaN)
  const int64_t a_ld = transb ? kp : np_, a_cols = transb ? np_ : kp;     // A' image: leading dimension x columns
  const int64_t b_ld = transa ? mp : kp, b_cols = transa ? kp : mp;
  const int64_t a_src_ld = transb ? k : n, a_src_cols = transb ? n : k;   // the host array behind it, as column-major ld x cols
  const int64_t b_src_ld = transa ? m : k, b_src_cols = transa ? k : m;
  const size_t ab = (size_t)a_ld * a_cols * sizeof(double), bb = (size_t)b_ld * b_cols * sizeof(double), cb = (size_t)np_ * mp * sizeof(double);
  void *pa = nullptr, *pb = nullptr, *pc = nullptr;
  if (pool_alloc(ctx, &pa, ab, nullptr) != 0) return -1;
  if (pool_alloc(ctx, &pb, bb, nullptr) != 0) { pool_free(ctx, pa, ab); return -1; }
  if (pool_alloc(ctx, &pc, cb, nullptr) != 0) { pool_free(ctx, pa, ab); pool_free(ctx, pb, bb); return -1; }
  hipStream_t st = ctx->s_main;
  int rc = 0;
  auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == 0) { rc = -1; set_error("lpgp_gemm_host: %s", hipGetErrorString(e)); } };
  ok(hipMemsetAsync(pa, 0, ab, st));
  ok(hipMemsetAsync(pb, 0, bb, st));
  ok(hipMemcpy2DAsync(pa, (size_t)a_ld * sizeof(double), B_host, (size_t)a_src_ld * sizeof(double), (size_t)a_src_ld * sizeof(double), (size_t)a_src_cols,
                      hipMemcpyHostToDevice, st));
  ok(hipMemcpy2DAsync(pb, (size_t)b_ld * sizeof(double), A_host, (size_t)b_src_ld * sizeof(double), (size_t)b_src_ld * sizeof(double), (size_t)b_src_cols,
                      hipMemcpyHostToDevice, st));
  if (beta != 0.0) {
    ok(hipMemsetAsync(pc, 0, cb, st));
    ok(hipMemcpy2DAsync(pc, (size_t)np_ * sizeof(double), C_host, (size_t)n * sizeof(double), (size_t)n * sizeof(double), (size_t)m, hipMemcpyHostToDevice, st));
  }
  if (rc == 0) {
    GemmArgs g;
    g.A = (const double*)pa; g.B = (const double*)pb; g.C = (double*)pc; g.lda = a_ld; g.ldb = b_ld; g.ldc = np_;
    g.mt = (int)(np_ / TILE); g.nt = (int)(mp / TILE); g.k = (int)kp; g.alpha = alpha; g.beta = beta;
    g.tri = 0;
    rc = launch_gemm(ctx, st, transb ? 1 : 0, transa ? 0 : 1, g, LPGP_K_GEMM);
  }
  if (rc == 0) ok(hipMemcpy2DAsync(C_host, (size_t)n * sizeof(double), pc, (size_t)np_ * sizeof(double), (size_t)n * sizeof(double), (size_t)m, hipMemcpyDeviceToHost, st));
  if (hipStreamSynchronize(st) != hipSuccess && rc == 0) { rc = -1; set_error("lpgp_gemm_host: synchronisation failed"); }
  pool_free(ctx, pa, ab);
  pool_free(ctx, pb, bb);
  pool_free(ctx, pc, cb);
  return rc;
}

int lpgp_rhs_to_host(lpgp_ctx* ctx, const lpgp_mat* mat, lpgp_rhs* rhs, double* out_host) {
  LPGP_CHECK(ctx && mat && rhs && out_host, "lpgp_rhs_to_host: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(rhs->ld == mat->pn, "lpgp_rhs_to_host: size mismatch");
  std::vector<double> h((size_t)rhs->ld * rhs->m);
  {
    int rc = rhs_clear_unassembled(ctx, mat, rhs);
    if (rc != 0) return rc;
  }
  LPGP_HIP(hipStreamSynchronize(ctx->s_main));       // assembly launches are asynchronous
  LPGP_HIP(hipMemcpy(h.data(), rhs->v, h.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (const auto& b : mat->blocks)
    for (int64_t i = 0; i < b.n; ++i)
      for (int64_t j = 0; j < rhs->m; ++j) out_host[(b.off + i) * rhs->m + j] = h[(size_t)(b.poff + i + j * rhs->ld)];
  return 0;
}

int lpgp_kernel_diag(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups, double* out_value) {
  LPGP_CHECK(kd && out_value, "lpgp_kernel_diag: null argument");
  (void)ctx;
  DevDesc desc;
  int rc = lower_kdesc(kd, ngroups, &desc);
  if (rc != 0) return rc;
  *out_value = desc_diag(desc);      // at x == x' every r_d = 0
  return 0;
}

int lpgp_kernel_matrix(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups, const lpgp_pts* X0,
                       const lpgp_pts* X1, double* out_host) {
  LPGP_CHECK(ctx && kd && X0 && X1 && out_host, "lpgp_kernel_matrix: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(X0->d == X1->d && kd[0].d == X0->d, "lpgp_kernel_matrix: dimension mismatch");
  if (X0->n == 0 || X1->n == 0) return 0;
  DevDesc desc;
  int rc = lower_kdesc(kd, ngroups, &desc);
  if (rc != 0) return rc;
  const int64_t ld = round_up(X0->n, 2);
  double* d = nullptr;
  LPGP_HIP(hipMalloc(&d, (size_t)ld * X1->n * sizeof(double)));
  rc = launch_assemble(ctx, ctx->s_main, desc, X0->x, X0->n, X0->n_pad, X1->x, X1->n, X1->n_pad, d, ld, 0, 0, 0);
  std::vector<double> h((size_t)ld * X1->n);
  if (rc == 0 && hipMemcpyAsync(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess) rc = -1;
  if (rc == 0 && hipStreamSynchronize(ctx->s_main) != hipSuccess) rc = -1;
  (void)hipFree(d);
  if (rc != 0) return rc;
  for (int64_t i = 0; i < X0->n; ++i)
    for (int64_t j = 0; j < X1->n; ++j) out_host[i * X1->n + j] = h[(size_t)(i + j * ld)];
  return 0;
}

// ---- measurement ----------------------------------------------------------------------------
int lpgp_kernel_matvec(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups, const lpgp_pts* X0,
                       const lpgp_pts* X1, const double* v_host, int64_t nrhs, double* out_host) {
  LPGP_CHECK(ctx && kd && X0 && X1 && v_host && out_host && nrhs >= 1, "lpgp_kernel_matvec: bad argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(X0->d == X1->d && kd[0].d == X0->d, "lpgp_kernel_matvec: dimension mismatch");
  const int64_t n0 = X0->n, n1 = X1->n;
  if (n0 == 0) return 0;
  if (n1 == 0) {
    std::memset(out_host, 0, (size_t)(n0 * nrhs) * sizeof(double));
    return 0;
  }
  DevDesc desc;
  int rc = lower_kdesc(kd, ngroups, &desc);
  if (rc != 0) return rc;
  // enough workgroups to fill the chip: row tiles x column splits >= ~4 per CU
  const int tiles_r = (int)((n0 + 63) / 64), tiles_c = (int)((n1 + 63) / 64);
  int splits = (4 * ctx->cus + tiles_r - 1) / tiles_r;
  if (splits > tiles_c) splits = tiles_c;
  if (splits < 1) splits = 1;
  const int64_t n0p = X0->n_pad, n1p = X1->n_pad;
  double *dv = nullptr, *dpart = nullptr, *dout = nullptr;
  void* p = nullptr;
  const size_t bv = (size_t)MV_RHS * n1p * sizeof(double), bp = (size_t)splits * MV_RHS * n0p * sizeof(double),
               bo = (size_t)MV_RHS * n0p * sizeof(double);
  if (pool_alloc(ctx, &p, bv + bp + bo, nullptr) != 0) return -1;
  dv = (double*)p;
  dpart = dv + (size_t)MV_RHS * n1p;
  dout = dpart + (size_t)splits * MV_RHS * n0p;
  std::vector<double> hv((size_t)MV_RHS * n1p), ho((size_t)MV_RHS * n0p);
  for (int64_t r0 = 0; r0 < nrhs && rc == 0; r0 += MV_RHS) {
    const int nr = (int)((nrhs - r0 < MV_RHS) ? nrhs - r0 : MV_RHS);
    for (int r = 0; r < nr; ++r)
      for (int64_t j = 0; j < n1; ++j) hv[(size_t)r * n1p + j] = v_host[j * nrhs + r0 + r];
    if (hipMemcpyAsync(dv, hv.data(), (size_t)nr * n1p * sizeof(double), hipMemcpyHostToDevice, ctx->s_main) != hipSuccess) rc = -1;
    if (rc == 0) rc = launch_matvec(ctx, ctx->s_main, desc, X0->x, n0, n0p, X1->x, n1, n1p, dv, nr, dpart, splits, dout);
    if (rc == 0 && hipMemcpyAsync(ho.data(), dout, (size_t)nr * n0p * sizeof(double), hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess) rc = -1;
    if (rc == 0 && hipStreamSynchronize(ctx->s_main) != hipSuccess) rc = -1;
    if (rc == 0)
      for (int r = 0; r < nr; ++r)
        for (int64_t i = 0; i < n0; ++i) out_host[i * nrhs + r0 + r] = ho[(size_t)r * n0p + i];
  }
  pool_free(ctx, p, bv + bp + bo);
  if (rc == -1) set_error("lpgp_kernel_matvec: HIP error %s", hipGetErrorString(hipGetLastError()));
  return rc;
}

int lpgp_profile_enable(lpgp_ctx* ctx, int32_t mask) {
  LPGP_DEVICE(ctx);
  int rc = prof_collect(ctx);
  ctx->prof_on = mask;
  return rc;
}

int lpgp_profile_reset(lpgp_ctx* ctx) {
  LPGP_DEVICE(ctx);
  int rc = prof_collect(ctx);
  for (auto& s : ctx->prof) s = ProfSlot();
  return rc;
}

int lpgp_profile_get(lpgp_ctx* ctx, int32_t kernel_id, double* ms, int64_t* launches, double* flops, double* bytes) {
  LPGP_CHECK(kernel_id >= 0 && kernel_id < LPGP_K_COUNT, "lpgp_profile_get: bad kernel id");
  LPGP_DEVICE(ctx);
  int rc = prof_collect(ctx);
  if (rc != 0) return rc;
  const ProfSlot& s = ctx->prof[kernel_id];
  if (ms) *ms = s.ms;
  if (launches) *launches = s.launches;
  if (flops) *flops = s.flops;
  if (bytes) *bytes = s.bytes;
  return 0;
}

}  // extern "C"
