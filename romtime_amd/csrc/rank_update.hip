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
