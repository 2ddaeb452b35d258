// Device-resident operands for the matrix-free path (round 6).
//
// The reference backs `CovarianceFunction.linop` / `_keops_lazy_tensor` with KeOps lazy tensors whose operands stay on the
// device (covfuncs/linfuncops/diffops/_matern.py:112-135, experiments/cpu.py:214-229), and probnum's `LinearOperator.solve`
// iterates on that product.  Until round 5 `lpgp_kernel_matvec` took and returned HOST vectors, so every iteration of the
// preconditioned conjugate gradients behind a matrix-free posterior (randprocs/_matrix_free.py) was upload + kernel + download
// + NumPy vector algebra -- and the rank-200 preconditioner, two host GEMVs over 52 MB at N = 32 768, cost more than the
// product.  Here:
//   lpgp_dvec                    a device-resident n x m block (column-major, one column per right-hand side)
//   lpgp_kernel_matvec_dev       Y[rows] (+)= K(X0, X1) V[rows] on such blocks: the matrix-free product with no host traffic
//   lpgp_dvec_scale_rows_add     Y[rows] += diag(d) V[rows]            (the noise term of a Gram block)
//   lpgp_pcg_*                   ONE iteration of preconditioned CG for all columns as launches only: column dots (fixed order
//                                of summation: two-stage, no atomics), step lengths formed on the device, the pivoted-Cholesky
//                                preconditioner M^{-1} = (I - L^T S^{-1} L) / delta applied as three small kernels.  The host
//                                reads back m residual norms per iteration (8 m bytes) to decide when to stop.

#include <cmath>
#include <cstring>
#include <vector>

#include "lpgp_internal.h"

struct lpgp_dvec {
  lpgp_ctx* ctx;
  int64_t n, m, ld;
  double* v;
  size_t bytes;
};

struct lpgp_pcg {
  lpgp_ctx* ctx;
  int64_t n, m;
  int32_t rank;
  double delta;
  double* L = nullptr;        // rank x n, row-major (pivot row k contiguous)
  double* Sinv = nullptr;     // rank x rank, symmetric: (delta I + L L^T)^{-1}
  double* work = nullptr;     // scalars and partial sums, see offsets below
  size_t L_bytes = 0, work_bytes = 0;
  int32_t G = 128;            // workgroups of a column dot (first stage)
  // work layout (doubles): part[3][m][G] | rz[m] | pq[m] | alpha[m] | beta[m] | rr[m] | bn[m] | rel[m] | active[m] | T[rank][m] | U[rank][m]
  __host__ __device__ double* part() const { return work; }
  __host__ __device__ double* scal(int k) const { return work + (size_t)3 * m * G + (size_t)k * m; }
  __host__ __device__ double* T() const { return work + (size_t)3 * m * G + (size_t)8 * m; }
  __host__ __device__ double* U() const { return T() + (size_t)rank * m; }
};

namespace lpgp {

constexpr int PCG_G = 128;

// part[c * G + g] = sum over the rows of workgroup g of A[i, c] * B[i, c]   (fixed assignment of rows: deterministic)
__global__ __launch_bounds__(256) void pcg_dot_kernel(const double* __restrict__ A, const double* __restrict__ B, int64_t n, int64_t ld, double* __restrict__ part) {
  __shared__ double red[256];
  const int c = blockIdx.y, g = blockIdx.x, G = gridDim.x;
  const double* a = A + (int64_t)c * ld;
  const double* b = B + (int64_t)c * ld;
  double s = 0.0;
  for (int64_t i = (int64_t)g * 256 + threadIdx.x; i < n; i += (int64_t)G * 256) s = fma(a[i], b[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(int64_t)c * G + g] = red[0];
}
__device__ __forceinline__ double pcg_sum(const double* part, int c, int G) {
  double s = 0.0;
  for (int g = 0; g < G; ++g) s += part[(int64_t)c * G + g];
  return s;
}
// which = 0 (start):   rz = <R, Z>, rr = <R, R>: rel = sqrt(rr) / bn, active = rel > rtol
// which = 1 (alpha):   pq = <P, Q>; alpha = active && pq > 0 ? rz / pq : 0
// which = 2 (beta):    rz' = <R, Z>, rr = <R, R>; beta = active_old ? rz' / rz : 0; rz = rz'; rel, active as above
__global__ void pcg_scalar_kernel(lpgp_pcg p, int which, double rtol) {
  const int c = threadIdx.x;
  if (c >= p.m) return;
  double *rz = p.scal(0), *pq = p.scal(1), *alpha = p.scal(2), *beta = p.scal(3), *rr = p.scal(4), *bn = p.scal(5), *rel = p.scal(6), *act = p.scal(7);
  const int G = p.G;
  if (which == 1) {
    const double v = pcg_sum(p.part(), c, G);
    pq[c] = v;
    alpha[c] = (act[c] != 0.0 && v > 0.0) ? rz[c] / v : 0.0;
    return;
  }
  const double rzn = pcg_sum(p.part() + (size_t)p.m * G, c, G), rrn = pcg_sum(p.part() + (size_t)2 * p.m * G, c, G);
  if (which == 2) beta[c] = (act[c] != 0.0 && rz[c] != 0.0) ? rzn / rz[c] : 0.0;
  rz[c] = rzn;
  rr[c] = rrn;
  rel[c] = sqrt(rrn) / bn[c];
  act[c] = rel[c] > rtol ? 1.0 : 0.0;
}
// X += alpha P;  R -= alpha Q
__global__ __launch_bounds__(256) void pcg_axpy2_kernel(double* __restrict__ X, double* __restrict__ R, const double* __restrict__ P, const double* __restrict__ Q,
                                                        int64_t n, int64_t ld, const double* __restrict__ alpha) {
  const int c = blockIdx.y;
  const double a = alpha[c];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t o = i + (int64_t)c * ld;
  X[o] = fma(a, P[o], X[o]);
  R[o] = fma(-a, Q[o], R[o]);
}
// P = Z + beta P
__global__ __launch_bounds__(256) void pcg_direction_kernel(double* __restrict__ P, const double* __restrict__ Z, int64_t n, int64_t ld, const double* __restrict__ beta) {
  const int c = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t o = i + (int64_t)c * ld;
  P[o] = fma(beta[c], P[o], Z[o]);
}
// T[k][c] = <L[k, :], R[:, c]>: one workgroup per pivot row and column
__global__ __launch_bounds__(256) void pcg_lr_kernel(const double* __restrict__ L, const double* __restrict__ R, int64_t n, int64_t ld, int m, double* __restrict__ T) {
  __shared__ double red[256];
  const int k = blockIdx.x, c = blockIdx.y;
  const double* l = L + (int64_t)k * n;
  const double* r = R + (int64_t)c * ld;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s = fma(l[i], r[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) T[(int64_t)k * m + c] = red[0];
}
// U = Sinv T   (rank x rank times rank x m)
__global__ void pcg_small_kernel(const double* __restrict__ Sinv, const double* __restrict__ T, int rank, int m, double* __restrict__ U) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (k >= rank) return;
  double s = 0.0;
  for (int j = 0; j < rank; ++j) s = fma(Sinv[(int64_t)k * rank + j], T[(int64_t)j * m + c], s);
  U[(int64_t)k * m + c] = s;
}
// Z = (R - L^T U) / delta   (rank == 0: Z = R / delta)
__global__ __launch_bounds__(256) void pcg_z_kernel(const double* __restrict__ L, const double* __restrict__ U, const double* __restrict__ R, double* __restrict__ Z,
                                                    int64_t n, int64_t ld, int rank, int m, double inv_delta) {
  const int c = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = 0; k < rank; ++k) s = fma(L[(int64_t)k * n + i], U[(int64_t)k * m + c], s);
  const int64_t o = i + (int64_t)c * ld;
  Z[o] = (R[o] - s) * inv_delta;
}
// Y[rows] += d[rows] * V[rows] for every column
__global__ __launch_bounds__(256) void dvec_scale_rows_add_kernel(double* __restrict__ Y, const double* __restrict__ V, const double* __restrict__ d, int64_t n, int64_t ld) {
  const int c = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Y[i + (int64_t)c * ld] = fma(d[i], V[i + (int64_t)c * ld], Y[i + (int64_t)c * ld]);
}
// out = a + s * b
__global__ __launch_bounds__(256) void dvec_axpby_kernel(double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ b, double s, int64_t n, int64_t ld) {
  const int c = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t o = i + (int64_t)c * ld;
  out[o] = fma(s, b[o], a[o]);
}

}  // namespace lpgp

using namespace lpgp;

extern "C" {

int lpgp_dvec_create(lpgp_ctx* ctx, int64_t n, int64_t m, lpgp_dvec** out) {
  LPGP_CHECK(ctx && out && n >= 1 && m >= 1, "lpgp_dvec_create: bad argument");
  LPGP_DEVICE(ctx);
  lpgp_dvec* d = new lpgp_dvec();
  d->ctx = ctx; d->n = n; d->m = m; d->ld = (n + 63) / 64 * 64;
  d->bytes = (size_t)d->ld * m * sizeof(double);
  void* p = nullptr;
  if (pool_alloc(ctx, &p, d->bytes, nullptr) != 0) { delete d; return -1; }
  d->v = (double*)p;
  if (hipMemsetAsync(d->v, 0, d->bytes, ctx->s_main) != hipSuccess) {
    (void)hipGetLastError();
    pool_free(ctx, p, d->bytes);
    delete d;
    LPGP_CHECK(false, "lpgp_dvec_create: clearing %zu bytes failed", (size_t)((n + 63) / 64 * 64) * (size_t)m * sizeof(double));
  }
  *out = d;
  return 0;
}
int lpgp_dvec_destroy(lpgp_dvec* d) {
  if (!d) return 0;
  (void)hipSetDevice(d->ctx->device);
  pool_free(d->ctx, d->v, d->bytes);
  delete d;
  return 0;
}
// host (n x m, C-order) -> device
int lpgp_dvec_set(lpgp_ctx* ctx, lpgp_dvec* d, const double* host) {
  LPGP_CHECK(ctx && d && host, "lpgp_dvec_set: null argument");
  LPGP_DEVICE(ctx);
  std::vector<double> h((size_t)d->ld * d->m, 0.0);
  for (int64_t i = 0; i < d->n; ++i)
    for (int64_t c = 0; c < d->m; ++c) h[(size_t)(i + c * d->ld)] = host[i * d->m + c];
  LPGP_HIP(hipMemcpyAsync(d->v, h.data(), d->bytes, hipMemcpyHostToDevice, ctx->s_main));
  LPGP_HIP(hipStreamSynchronize(ctx->s_main));
  return 0;
}
int lpgp_dvec_get(lpgp_ctx* ctx, const lpgp_dvec* d, double* host) {
  LPGP_CHECK(ctx && d && host, "lpgp_dvec_get: null argument");
  LPGP_DEVICE(ctx);
  std::vector<double> h((size_t)d->ld * d->m);
  LPGP_HIP(hipMemcpyAsync(h.data(), d->v, d->bytes, hipMemcpyDeviceToHost, ctx->s_main));
  LPGP_HIP(hipStreamSynchronize(ctx->s_main));
  for (int64_t i = 0; i < d->n; ++i)
    for (int64_t c = 0; c < d->m; ++c) host[i * d->m + c] = h[(size_t)(i + c * d->ld)];
  return 0;
}
// out = a + s * b (all of one shape; out may alias a or b)
int lpgp_dvec_axpby(lpgp_ctx* ctx, lpgp_dvec* out, const lpgp_dvec* a, const lpgp_dvec* b, double s) {
  LPGP_CHECK(ctx && out && a && b && out->n == a->n && a->n == b->n && out->m == a->m && a->m == b->m, "lpgp_dvec_axpby: shape mismatch");
  LPGP_DEVICE(ctx);
  hipLaunchKernelGGL(dvec_axpby_kernel, dim3((unsigned)((a->n + 255) / 256), (unsigned)a->m), dim3(256), 0, ctx->s_main, out->v, a->v, b->v, s, a->n, a->ld);
  LPGP_HIP(hipGetLastError());
  return 0;
}
// Y[y_off : y_off + n, :] += diag(d_host) V[v_off : v_off + n, :]   (d_host: n values; the noise diagonal of a Gram block)
int lpgp_dvec_scale_rows_add(lpgp_ctx* ctx, lpgp_dvec* Y, int64_t y_off, const lpgp_dvec* V, int64_t v_off, const double* d_host, int64_t n) {
  LPGP_CHECK(ctx && Y && V && d_host && n >= 1 && Y->m == V->m && Y->ld == V->ld && y_off == v_off && y_off >= 0 && y_off + n <= Y->n,
             "lpgp_dvec_scale_rows_add: bad argument (the two blocks must share their layout and row range)");
  LPGP_DEVICE(ctx);
  void* p = nullptr;
  const size_t b = (size_t)n * sizeof(double);
  if (pool_alloc(ctx, &p, b, nullptr) != 0) return -1;
  hipError_t e = hipMemcpyAsync(p, d_host, b, hipMemcpyHostToDevice, ctx->s_main);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(dvec_scale_rows_add_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)Y->m), dim3(256), 0, ctx->s_main, Y->v + y_off, V->v + v_off,
                       (const double*)p, n, Y->ld);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->s_main);          // (borrowed host vector)
  pool_free(ctx, p, b);
  LPGP_CHECK(e == hipSuccess, "lpgp_dvec_scale_rows_add: %s", hipGetErrorString(e));
  return 0;
}

// Y[y_off : y_off + n0, :] (accumulate: +=) K(X0, X1) V[v_off : v_off + n1, :], every operand resident (replaces the KeOps lazy
// product whose operands stay on the device, diffops/_matern.py:112-135).  No host synchronisation.
int lpgp_kernel_matvec_dev(lpgp_ctx* ctx, const lpgp_kdesc* kd, int32_t ngroups, const lpgp_pts* X0, const lpgp_pts* X1, const lpgp_dvec* V, int64_t v_off,
                           lpgp_dvec* Y, int64_t y_off, int32_t accumulate) {
  LPGP_CHECK(ctx && kd && X0 && X1 && V && Y, "lpgp_kernel_matvec_dev: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(X0->d == X1->d && kd[0].d == X0->d, "lpgp_kernel_matvec_dev: dimension mismatch");
  const int64_t n0 = X0->n, n1 = X1->n;
  LPGP_CHECK(V->m == Y->m && v_off >= 0 && v_off + n1 <= V->n && y_off >= 0 && y_off + n0 <= Y->n, "lpgp_kernel_matvec_dev: block out of range");
  if (n0 == 0 || n1 == 0) return 0;
  DevDesc desc;
  int rc = lower_kdesc(kd, ngroups, &desc);
  if (rc != 0) return rc;
  const int tiles_r = (int)((n0 + 63) / 64), tiles_c = (int)((n1 + 63) / 64);
  int splits = (4 * ctx->cus + tiles_r - 1) / tiles_r;
  if (splits > tiles_c) splits = tiles_c;
  if (splits < 1) splits = 1;
  const int64_t n0p = X0->n_pad;
  void* p = nullptr;
  const size_t bp = (size_t)splits * MV_RHS * n0p * sizeof(double);
  if (pool_alloc(ctx, &p, bp, nullptr) != 0) return -1;
  for (int64_t r0 = 0; r0 < V->m && rc == 0; r0 += MV_RHS) {
    const int nr = (int)((V->m - r0 < MV_RHS) ? V->m - r0 : MV_RHS);
    rc = launch_matvec(ctx, ctx->s_main, desc, X0->x, n0, n0p, X1->x, n1, X1->n_pad, V->v + v_off + r0 * V->ld, nr, (double*)p, splits,
                       Y->v + y_off + r0 * Y->ld, V->ld, Y->ld, accumulate);
  }
  pool_free(ctx, p, bp);         // (reuse is stream-ordered: every user of the pool runs on the panel stream)
  return rc;
}

// ---- preconditioned conjugate gradients, one iteration = launches only ------------------------------------------------
int lpgp_pcg_create(lpgp_ctx* ctx, int64_t n, int64_t m, int32_t rank, const double* L_host /* rank x n */, const double* Sinv_host /* rank x rank */,
                    double delta, lpgp_pcg** out) {
  LPGP_CHECK(ctx && out && n >= 1 && m >= 1 && m <= 256 && rank >= 0 && delta > 0.0 && (rank == 0 || (L_host && Sinv_host)), "lpgp_pcg_create: bad argument");
  LPGP_DEVICE(ctx);
  lpgp_pcg* p = new lpgp_pcg();
  p->ctx = ctx; p->n = n; p->m = m; p->rank = rank; p->delta = delta; p->G = PCG_G;
  p->L_bytes = ((size_t)rank * n + (size_t)rank * rank + 8) * sizeof(double);
  p->work_bytes = ((size_t)3 * m * PCG_G + (size_t)8 * m + (size_t)2 * rank * m + 8) * sizeof(double);
  void *pl = nullptr, *pw = nullptr;
  if (pool_alloc(ctx, &pl, p->L_bytes, nullptr) != 0) { delete p; return -1; }
  if (pool_alloc(ctx, &pw, p->work_bytes, nullptr) != 0) { pool_free(ctx, pl, p->L_bytes); delete p; return -1; }
  p->L = (double*)pl;
  p->Sinv = p->L + (size_t)rank * n;
  p->work = (double*)pw;
  hipError_t e = hipMemsetAsync(pw, 0, p->work_bytes, ctx->s_main);
  if (e == hipSuccess && rank > 0) e = hipMemcpyAsync(p->L, L_host, (size_t)rank * n * sizeof(double), hipMemcpyHostToDevice, ctx->s_main);
  if (e == hipSuccess && rank > 0) e = hipMemcpyAsync(p->Sinv, Sinv_host, (size_t)rank * rank * sizeof(double), hipMemcpyHostToDevice, ctx->s_main);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->s_main);
  if (e != hipSuccess) {
    set_error("lpgp_pcg_create: %s", hipGetErrorString(e));
    pool_free(ctx, pl, p->L_bytes); pool_free(ctx, pw, p->work_bytes);
    delete p;
    return -1;
  }
  *out = p;
  return 0;
}
int lpgp_pcg_destroy(lpgp_pcg* p) {
  if (!p) return 0;
  (void)hipSetDevice(p->ctx->device);
  pool_free(p->ctx, p->L, p->L_bytes);
  pool_free(p->ctx, p->work, p->work_bytes);
  delete p;
  return 0;
}
static int pcg_precond(lpgp_pcg* p, const lpgp_dvec* R, lpgp_dvec* Z) {
  hipStream_t st = p->ctx->s_main;
  const int m = (int)p->m;
  if (p->rank > 0) {
    hipLaunchKernelGGL(pcg_lr_kernel, dim3((unsigned)p->rank, (unsigned)m), dim3(256), 0, st, (const double*)p->L, (const double*)R->v, p->n, R->ld, m, p->T());
    hipLaunchKernelGGL(pcg_small_kernel, dim3((unsigned)((p->rank + 63) / 64), (unsigned)m), dim3(64), 0, st, (const double*)p->Sinv, (const double*)p->T(), p->rank, m, p->U());
  }
  hipLaunchKernelGGL(pcg_z_kernel, dim3((unsigned)((p->n + 255) / 256), (unsigned)m), dim3(256), 0, st, (const double*)p->L, (const double*)p->U(), (const double*)R->v, Z->v,
                     p->n, R->ld, p->rank, m, 1.0 / p->delta);
  LPGP_HIP(hipGetLastError());
  return 0;
}
static int pcg_dots_rz_rr(lpgp_pcg* p, const lpgp_dvec* R, const lpgp_dvec* Z) {
  hipStream_t st = p->ctx->s_main;
  hipLaunchKernelGGL(pcg_dot_kernel, dim3(PCG_G, (unsigned)p->m), dim3(256), 0, st, (const double*)R->v, (const double*)Z->v, p->n, R->ld, p->part() + (size_t)p->m * PCG_G);
  hipLaunchKernelGGL(pcg_dot_kernel, dim3(PCG_G, (unsigned)p->m), dim3(256), 0, st, (const double*)R->v, (const double*)R->v, p->n, R->ld, p->part() + (size_t)2 * p->m * PCG_G);
  LPGP_HIP(hipGetLastError());
  return 0;
}
// start: R holds B - G X0 (the caller formed it), bnorm_host[m] the norms the residuals are measured against; Z = M^{-1} R,
// P = Z, rz = <R, Z>; rel_host[m] <- ||R|| / bnorm (one read-back)
int lpgp_pcg_start(lpgp_ctx* ctx, lpgp_pcg* p, const lpgp_dvec* R, lpgp_dvec* Z, lpgp_dvec* P, const double* bnorm_host, double rtol, double* rel_host) {
  LPGP_CHECK(ctx && p && R && Z && P && bnorm_host && rel_host, "lpgp_pcg_start: null argument");
  LPGP_DEVICE(ctx);
  LPGP_CHECK(R->n == p->n && R->m == p->m && Z->ld == R->ld && P->ld == R->ld && Z->m == R->m && P->m == R->m, "lpgp_pcg_start: shape mismatch");
  hipStream_t st = ctx->s_main;
  LPGP_HIP(hipMemcpyAsync(p->scal(5), bnorm_host, (size_t)p->m * sizeof(double), hipMemcpyHostToDevice, st));
  LPGP_HIP(hipStreamSynchronize(st));
  LPGP_TRY(pcg_precond(p, R, Z));
  LPGP_HIP(hipMemcpyAsync(P->v, Z->v, Z->bytes, hipMemcpyDeviceToDevice, st));
  LPGP_TRY(pcg_dots_rz_rr(p, R, Z));
  hipLaunchKernelGGL(pcg_scalar_kernel, dim3(1), dim3(256), 0, st, *p, 0, rtol);
  LPGP_HIP(hipGetLastError());
  LPGP_HIP(hipMemcpyAsync(rel_host, p->scal(6), (size_t)p->m * sizeof(double), hipMemcpyDeviceToHost, st));
  LPGP_HIP(hipStreamSynchronize(st));
  return 0;
}
// one iteration AFTER the caller has formed Q = G P (lpgp_kernel_matvec_dev ...): step lengths, X += alpha P, R -= alpha Q,
// Z = M^{-1} R, beta, P = Z + beta P -- eleven launches, no host arithmetic; rel_host[m] <- ||R|| / bnorm
int lpgp_pcg_step(lpgp_ctx* ctx, lpgp_pcg* p, lpgp_dvec* X, lpgp_dvec* R, lpgp_dvec* Z, lpgp_dvec* P, const lpgp_dvec* Q, double rtol, double* rel_host) {
  LPGP_CHECK(ctx && p && X && R && Z && P && Q && rel_host, "lpgp_pcg_step: null argument");
  LPGP_DEVICE(ctx);
  for (const lpgp_dvec* d : {(const lpgp_dvec*)X, (const lpgp_dvec*)R, (const lpgp_dvec*)Z, (const lpgp_dvec*)P, Q})
    LPGP_CHECK(d->n == p->n && d->m == p->m, "lpgp_pcg_step: a vector block of %lld x %lld where the iteration was created for %lld x %lld", (long long)d->n,
               (long long)d->m, (long long)p->n, (long long)p->m);
  hipStream_t st = ctx->s_main;
  const dim3 gv((unsigned)((p->n + 255) / 256), (unsigned)p->m);
  hipLaunchKernelGGL(pcg_dot_kernel, dim3(PCG_G, (unsigned)p->m), dim3(256), 0, st, (const double*)P->v, (const double*)Q->v, p->n, P->ld, p->part());
  hipLaunchKernelGGL(pcg_scalar_kernel, dim3(1), dim3(256), 0, st, *p, 1, rtol);
  hipLaunchKernelGGL(pcg_axpy2_kernel, gv, dim3(256), 0, st, X->v, R->v, (const double*)P->v, (const double*)Q->v, p->n, X->ld, (const double*)p->scal(2));
  LPGP_HIP(hipGetLastError());
  LPGP_TRY(pcg_precond(p, R, Z));
  LPGP_TRY(pcg_dots_rz_rr(p, R, Z));
  hipLaunchKernelGGL(pcg_scalar_kernel, dim3(1), dim3(256), 0, st, *p, 2, rtol);
  hipLaunchKernelGGL(pcg_direction_kernel, gv, dim3(256), 0, st, P->v, (const double*)Z->v, p->n, P->ld, (const double*)p->scal(3));
  LPGP_HIP(hipGetLastError());
  LPGP_HIP(hipMemcpyAsync(rel_host, p->scal(6), (size_t)p->m * sizeof(double), hipMemcpyDeviceToHost, st));
  LPGP_HIP(hipStreamSynchronize(st));
  return 0;
}

}  // extern "C"
