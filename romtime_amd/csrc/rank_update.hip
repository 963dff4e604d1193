// Thin update of a tall matrix:  Y_dst (N x n) = Y_src diag(d) + alpha X (N x k) T (k x n),  k <= 64, all row-major.
// The deflation sweep of the levelled POD (pod.py: X <- X - Q (Q^T X), SURVEY 8a1 hard part A) is this with
// alpha = -1: 16 N n bytes for 2 N n k flops, i.e. HBM-bound for the few modes a level accepts.  The generic GEMM
// reached 2.2 TB/s on it (its tiles are shaped for a long contraction); here a workgroup owns a 128-column strip,
// keeps T's strip in LDS for its whole life, stages the X rows of 32 rows at a time, and every thread updates two
// adjacent columns of eight rows with its loads issued before the first FMA.  Out of place (Y_src != Y_dst) it also
// replaces the clone of the caller's snapshots and, with d, their column normalisation.
#include "common.h"

typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int RU_THREADS = 256;
constexpr int RU_COLS = 128;   // columns per strip (two per thread of a 64-thread row group)
constexpr int RU_ROWS = 32;    // rows per stage (four row groups x eight rows)
constexpr int RU_KMAX = 64;

struct RuParams {
  const double* Ysrc;
  const double* X;
  const double* T;
  const double* d;   // optional column scale of Y_src (n), nullptr = 1
  double* Ydst;
  long N, ldys, ldx, ldt, ldyd;
  int n, k;
  double alpha;
};

__global__ __launch_bounds__(RU_THREADS) void rank_update_kernel(const RuParams p) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* sT = sm;                       // [k][RU_COLS]
  double* sX = sm + (size_t)p.k * RU_COLS;  // [RU_ROWS][k]
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int col0 = blockIdx.y * RU_COLS, c = col0 + 2 * tx;
  const int k = p.k;
  for (int q = tid; q < k * RU_COLS; q += RU_THREADS) {
    const int j = q / RU_COLS, cc = col0 + q % RU_COLS;
    sT[q] = (cc < p.n) ? p.alpha * p.T[(long)j * p.ldt + cc] : 0.0;
  }
  const bool vec = ((p.ldys | p.ldyd) & 1) == 0 && ((reinterpret_cast<size_t>(p.Ysrc) | reinterpret_cast<size_t>(p.Ydst)) & 15) == 0 &&
                   c + 1 < p.n;
  const bool have0 = c < p.n, have1 = c + 1 < p.n;
  d2 scale{1.0, 1.0};
  if (p.d) {
    if (have0) scale.x = p.d[c];
    if (have1) scale.y = p.d[c + 1];
  }
  for (long row0 = (long)blockIdx.x * RU_ROWS; row0 < p.N; row0 += (long)gridDim.x * RU_ROWS) {
    __syncthreads();  // the previous stage's sX is no longer read (and sT is complete)
    for (int q = tid; q < RU_ROWS * k; q += RU_THREADS) {
      const long row = row0 + q / k;
      sX[q] = (row < p.N) ? p.X[row * p.ldx + q % k] : 0.0;
    }
    d2 y[RU_ROWS / 4];
#pragma unroll
    for (int i = 0; i < RU_ROWS / 4; ++i) {  // all loads of the stage in flight before the first FMA
      const long row = row0 + ty + 4 * i;
      d2 v{0.0, 0.0};
      if (row < p.N) {
        const double* src = p.Ysrc + row * p.ldys + c;
        if (vec) {
          v = *reinterpret_cast<const d2*>(src);
        } else {
          if (have0) v.x = src[0];
          if (have1) v.y = src[1];
        }
      }
      y[i] = v * scale;
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {
      const d2 t = *reinterpret_cast<const d2*>(&sT[j * RU_COLS + 2 * tx]);
#pragma unroll
      for (int i = 0; i < RU_ROWS / 4; ++i) {
        const double x = sX[(ty + 4 * i) * k + j];
        y[i].x = fma(x, t.x, y[i].x);
        y[i].y = fma(x, t.y, y[i].y);
      }
    }
#pragma unroll
    for (int i = 0; i < RU_ROWS / 4; ++i) {
      const long row = row0 + ty + 4 * i;
      if (row < p.N) {
        double* dst = p.Ydst + row * p.ldyd + c;
        if (vec) {
          *reinterpret_cast<d2*>(dst) = y[i];
        } else {
          if (have0) dst[0] = y[i].x;
          if (have1) dst[1] = y[i].y;
        }
      }
    }
  }
}

}  // namespace

extern "C" int rt_rank_update(rt_ctx* ctx, const double* Ysrc, int64_t ldys, const double* colscale, const double* X,
                              int64_t ldx, const double* T, int64_t ldt, int64_t N, int64_t k, int64_t n, double alpha,
                              double* Ydst, int64_t ldyd) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, Ysrc && X && T && Ydst && N >= 1 && k >= 1 && n >= 1 && ldys >= n && ldyd >= n && ldx >= k && ldt >= n);
  if (k > RU_KMAX) {
    ctx->err = "rt_rank_update: k > 64 (use rt_gemm_nn_axpby)";
    return RT_ERR_UNSUPPORTED;
  }
  RuParams p{Ysrc, X, T, colscale, Ydst, (long)N, (long)ldys, (long)ldx, (long)ldt, (long)ldyd, (int)n, (int)k, alpha};
  const size_t lds = sizeof(double) * ((size_t)k * RU_COLS + (size_t)RU_ROWS * k);
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&rank_update_kernel),
                     (int)(sizeof(double) * RU_KMAX * (RU_COLS + RU_ROWS))));
  const long stages = (N + RU_ROWS - 1) / RU_ROWS;
  const unsigned gy = (unsigned)((n + RU_COLS - 1) / RU_COLS);
  long gx = (long)ctx->num_cus * 8 / gy;
  if (gx < 1) gx = 1;
  if (gx > stages) gx = stages;
  hipLaunchKernelGGL(rank_update_kernel, dim3((unsigned)gx, gy), dim3(RU_THREADS), lds, ctx->stream, p);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The other half of the deflation sweep:  C (m x n) = A^T B  with A (N x m) the few accepted modes and B (N x n) the
// working copy, both row-major, N long.  8 N n bytes for 2 N n m flops: with m <= 16 it is a stream over B.  The generic
// GEMM (tiles shaped for a long contraction and a wide output) reached 2.4 TB/s on it.  Here a thread owns two adjacent
// columns of B and all m rows of C in registers; a workgroup walks a contiguous range of rows, eight rows of loads in
// flight before the first FMA, the m entries of A's row through the scalar cache (their address is uniform).  Every
// workgroup leaves its partial C in a slab; a second kernel adds the slabs in a fixed order (no atomics: the result does
// not depend on scheduling).
namespace {

constexpr int ST_MMAX = 16;
constexpr int ST_ROWS = 8;     // rows of loads in flight per thread
constexpr int ST_GLANES = 16;  // slabs summed side by side in the reduction

template <int M>
__global__ __launch_bounds__(256) void skinny_tn_kernel(const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                        long ldb, long N, int n, long rows_per_wg,
                                                        double* __restrict__ slab) {
  const int c = 2 * (blockIdx.y * blockDim.x + threadIdx.x);
  const long r0 = (long)blockIdx.x * rows_per_wg;
  const long r1 = (r0 + rows_per_wg < N) ? r0 + rows_per_wg : N;
  const bool have0 = c < n, have1 = c + 1 < n;
  const bool vec = (ldb & 1) == 0 && (reinterpret_cast<size_t>(B) & 15) == 0 && have1;
  d2 acc[M];
#pragma unroll
  for (int i = 0; i < M; ++i) acc[i] = d2{0.0, 0.0};
  for (long r = r0; r < r1; r += ST_ROWS) {
    d2 x[ST_ROWS];
#pragma unroll
    for (int u = 0; u < ST_ROWS; ++u) {
      const long row = r + u;
      d2 v{0.0, 0.0};
      if (row < r1) {
        const double* src = B + row * ldb + c;
        if (vec) {
          v = *reinterpret_cast<const d2*>(src);
        } else {
          if (have0) v.x = src[0];
          if (have1) v.y = src[1];
        }
      }
      x[u] = v;
    }
#pragma unroll
    for (int u = 0; u < ST_ROWS; ++u) {
      const long row = (r + u < r1) ? r + u : r1 - 1;   // past the end x[u] is zero; keep A's address in range
      const double* __restrict__ a = A + row * lda;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const double ai = a[i];
        acc[i].x = fma(ai, x[u].x, acc[i].x);
        acc[i].y = fma(ai, x[u].y, acc[i].y);
      }
    }
  }
  double* dst = slab + (size_t)blockIdx.x * M * n;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    if (have0) dst[(size_t)i * n + c] = acc[i].x;
    if (have1) dst[(size_t)i * n + c + 1] = acc[i].y;
  }
}

__global__ __launch_bounds__(64 * ST_GLANES) void skinny_tn_reduce_kernel(const double* __restrict__ slab, int slabs, long mn,
                                                                         int n, double* __restrict__ Cm, long ldc) {
  __shared__ double part[ST_GLANES][64];
  const int lane = threadIdx.x & 63, gl = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + lane;
  double s = 0.0;
  if (e < mn)
    for (int g = gl; g < slabs; g += ST_GLANES) s += slab[(size_t)g * mn + e];
  part[gl][lane] = s;
  __syncthreads();
  if (gl == 0 && e < mn) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < ST_GLANES; ++q) t += part[q][lane];
    Cm[(e / n) * ldc + e % n] = t;
  }
}

template <int M>
void launch_skinny_tn(hipStream_t st, dim3 grid, int threads, const double* A, long lda, const double* B, long ldb, long N,
                      int n, long rows_per_wg, double* slab) {
  hipLaunchKernelGGL(skinny_tn_kernel<M>, grid, dim3(threads), 0, st, A, lda, B, ldb, N, n, rows_per_wg, slab);
}

}  // namespace

// C (m x n, row-major, ldc) = A^T B for row-major A (N x m), B (N x n), m <= 16, N long; RT_ERR_UNSUPPORTED otherwise.
int rt_skinny_tn(rt_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, int64_t N, int64_t m, int64_t n,
                 double* Cm, int64_t ldc) {
  static const int flags = [] { const char* e = getenv("ROMTIME_DEFLATE_FLAGS"); return e ? atoi(e) : 0; }();
  if ((flags & 1) || m > ST_MMAX || n > 4096 || N < 16384 || N * n < (1L << 22)) return RT_ERR_UNSUPPORTED;
  const int pairs = (int)((n + 1) / 2);
  const int threads = pairs >= 256 ? 256 : ((pairs + 63) / 64) * 64;
  const unsigned gy = (unsigned)((pairs + threads - 1) / threads);
  const long waves = (long)(threads / 64) * gy;
  // four waves per SIMD (the register budget of m = 16 allows no more): 5.7 TB/s on 1e6 x 512 with m = 8 against 4.7 with two;
  // but no more slabs than a sixteenth of B's bytes
  long gx = (long)ctx->num_cus * ((flags & 2) ? 8 : (flags & 4) ? 32 : 16) / waves;
  if (gx > N / (16 * m)) gx = N / (16 * m);
  if (gx < 1) gx = 1;
  long rows_per_wg = (N + gx - 1) / gx;
  rows_per_wg = (rows_per_wg + ST_ROWS - 1) / ST_ROWS * ST_ROWS;
  gx = (N + rows_per_wg - 1) / rows_per_wg;
  void* slab = nullptr;
  RT_TRY(rt_scratch(ctx, sizeof(double) * (size_t)gx * m * n, &slab));
  const dim3 grid((unsigned)gx, gy);
  using launch_fn = void (*)(hipStream_t, dim3, int, const double*, long, const double*, long, long, int, long, double*);
  static const launch_fn table[ST_MMAX] = {
      launch_skinny_tn<1>,  launch_skinny_tn<2>,  launch_skinny_tn<3>,  launch_skinny_tn<4>,
      launch_skinny_tn<5>,  launch_skinny_tn<6>,  launch_skinny_tn<7>,  launch_skinny_tn<8>,
      launch_skinny_tn<9>,  launch_skinny_tn<10>, launch_skinny_tn<11>, launch_skinny_tn<12>,
      launch_skinny_tn<13>, launch_skinny_tn<14>, launch_skinny_tn<15>, launch_skinny_tn<16>};
  table[m - 1](ctx->stream, grid, threads, A, (long)lda, B, (long)ldb, (long)N, (int)n, rows_per_wg, (double*)slab);
  const long mn = (long)m * n;
  hipLaunchKernelGGL(skinny_tn_reduce_kernel, dim3((unsigned)((mn + 63) / 64)), dim3(64 * ST_GLANES), 0, ctx->stream,
                     (const double*)slab, (int)gx, mn, (int)n, Cm, (long)ldc);
  ctx->last_grid = gx * gy; ctx->last_splits = gx; ctx->last_tile = (int)m * 1000 + 2 * threads;
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}
