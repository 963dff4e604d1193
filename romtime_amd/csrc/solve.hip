// Batched small dense solve: LU with partial pivoting, one workgroup per system, the whole
// r x r matrix resident in LDS (r <= 128 -> 128 KiB of the CU's 160 KiB).
// Stands in for np.linalg.solve (theta solve, deim.py:491-492) and for the GMRES(20) call on
// the dense reduced system (rom.py:36,492): a direct solve meets the reference's 1e-10
// residual target with margin (SURVEY.md hard part E).
#include "common.h"

namespace {

constexpr int SOLVE_THREADS = 256;

__global__ __launch_bounds__(SOLVE_THREADS) void dense_solve_kernel(double* __restrict__ K, double* __restrict__ rhs,
                                                                    int r, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lda = r | 1;  // odd leading dimension: column walks hit distinct banks
  double* A = sm;         // r x lda
  double* b = sm + (size_t)r * lda;
  __shared__ int s_piv;
  __shared__ int s_sing;
  const int tid = threadIdx.x;
  double* Kb = K + (size_t)blockIdx.x * r * r;
  double* rb = rhs + (size_t)blockIdx.x * r;
  for (int e = tid; e < r * r; e += SOLVE_THREADS) A[(e / r) * lda + (e % r)] = Kb[e];
  for (int e = tid; e < r; e += SOLVE_THREADS) b[e] = rb[e];
  if (tid == 0) s_sing = 0;
  __syncthreads();

  for (int c = 0; c < r; ++c) {
    // pivot search by the first wave: argmax |A[i][c]|, i >= c, lowest i on ties (LAPACK idamax)
    if (tid < 64) {
      double best = -1.0;
      int bi = c;
      for (int i = c + tid; i < r; i += 64) {
        const double v = fabs(A[i * lda + c]);
        if (v > best) {
          best = v;
          bi = i;
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off);
        const int oi = __shfl_down(bi, off);
        if (ov > best || (ov == best && oi < bi)) {
          best = ov;
          bi = oi;
        }
      }
      if (tid == 0) {
        s_piv = bi;
        if (best == 0.0) s_sing = 1;
      }
    }
    __syncthreads();
    const int pr = s_piv;
    if (pr != c) {
      for (int j = tid; j < r; j += SOLVE_THREADS) {
        const double t = A[c * lda + j];
        A[c * lda + j] = A[pr * lda + j];
        A[pr * lda + j] = t;
      }
      if (tid == 0) {
        const double t = b[c];
        b[c] = b[pr];
        b[pr] = t;
      }
    }
    __syncthreads();
    const double inv = 1.0 / A[c * lda + c];
    // multipliers + rank-1 update of the trailing block; thread -> (row, column-chunk)
    const int nrow = r - c - 1;
    for (int e = tid; e < nrow * 4; e += SOLVE_THREADS) {
      const int i = c + 1 + e / 4, part = e % 4;
      const double l = A[i * lda + c] * inv;
      for (int j = c + 1 + part; j < r; j += 4) A[i * lda + j] = fma(-l, A[c * lda + j], A[i * lda + j]);
      if (part == 0) b[i] = fma(-l, b[c], b[i]);
    }
    __syncthreads();
    for (int i = c + 1 + tid; i < r; i += SOLVE_THREADS) A[i * lda + c] *= inv;  // store L
    __syncthreads();
  }
  // back substitution U x = b (column oriented)
  for (int c = r - 1; c >= 0; --c) {
    if (tid == 0) b[c] = b[c] / A[c * lda + c];
    __syncthreads();
    const double xc = b[c];
    for (int i = tid; i < c; i += SOLVE_THREADS) b[i] = fma(-A[i * lda + c], xc, b[i]);
    __syncthreads();
  }
  for (int e = tid; e < r * r; e += SOLVE_THREADS) Kb[e] = A[(e / r) * lda + (e % r)];
  for (int e = tid; e < r; e += SOLVE_THREADS) rb[e] = b[e];
  if (info && tid == 0) info[blockIdx.x] = s_sing ? RT_WARN_SINGULAR : 0;
}

}  // namespace

extern "C" int rt_dense_solve_batched(rt_ctx* ctx, double* K, double* rhs, int64_t r, int64_t B, int* info) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, K && rhs && r >= 1 && B >= 1);
  if (r > 128) {
    ctx->err = "rt_dense_solve_batched: r > 128 not supported (matrix must fit the CU's LDS)";
    return RT_ERR_UNSUPPORTED;
  }
  const int lda = (int)r | 1;
  const size_t lds = sizeof(double) * ((size_t)r * lda + r);
  static bool attr_set = false;
  if (!attr_set) {
    RT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_solve_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(dense_solve_kernel, dim3((unsigned)B), dim3(SOLVE_THREADS), lds, ctx->stream, K, rhs, (int)r,
                     info);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}
