// Batched small dense solve: LU with partial pivoting, one workgroup per system, the whole
// r x r matrix resident in LDS (r <= 128 -> 128 KiB of the CU's 160 KiB).
// Stands in for np.linalg.solve (theta solve, deim.py:491-492) and for the GMRES(20) call on
// the dense reduced system (rom.py:36,492): a direct solve meets the reference's 1e-10
// residual target with margin (SURVEY.md hard part E).
//
// Latency-bound, so the elimination is organised around ONE workgroup barrier per column:
//   * rows are never swapped physically: a row that has served as pivot is retired (implicit permutation),
//   * thread (row, part) owns the columns j == part (mod parts) of its row for the whole factorisation, the
//     right-hand side rides along, and the multipliers are not stored (the factors are not an output),
//   * while a thread updates column c+1 of its row it already knows that row's candidate for the next pivot
//     search; the per-wave maxima (DPP reduction, no LDS shuffles) go to a double-buffered LDS slot, and after
//     the barrier every thread finishes the argmax redundantly instead of waiting for one wave to do it,
//   * the LDS reads of a batch of columns are all issued before the first write of the batch,
//   * back substitution runs in the first wave alone (no barriers).
#include "common.h"
#include "wave_ops.h"

namespace {

constexpr int SOLVE_THREADS = 256;
constexpr int SOLVE_WAVES = SOLVE_THREADS / 64;
constexpr int SOLVE_BATCH = 4;

__global__ __launch_bounds__(SOLVE_THREADS) void dense_solve_kernel(const double* __restrict__ K,
                                                                    double* __restrict__ rhs, int r, int parts,
                                                                    int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lda = r | 1;  // odd leading dimension: column walks hit distinct banks
  double* A = sm;         // r x lda
  double* b = sm + (size_t)r * lda;
  __shared__ double s_pv[2][SOLVE_WAVES];
  __shared__ int s_pi[2][SOLVE_WAVES];
  __shared__ int s_perm[128];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double* Kb = K + (size_t)blockIdx.x * r * r;
  double* rb = rhs + (size_t)blockIdx.x * r;
  for (int i = wid; i < r; i += SOLVE_WAVES)
    for (int j = lane; j < r; j += 64) A[i * lda + j] = Kb[i * r + j];
  for (int e = tid; e < r; e += SOLVE_THREADS) b[e] = rb[e];
  __syncthreads();

  const int row = tid / parts, part = tid - row * parts;
  const bool owner = row < r;
  bool retired = !owner;
  double* Ar = A + (owner ? row : 0) * lda;
  int sing = 0;
  int cm = 1 % parts;  // (c + 1) mod parts

  double cand = (owner && part == 0) ? fabs(Ar[0]) : -1.0;  // candidates of column 0
  for (int c = 0; c < r; ++c) {
    int ci = row;
    rtw::wave_argmax(cand, ci);  // (|value| desc, row asc) over the wave's live rows
    if (lane == 0) {
      s_pv[c & 1][wid] = cand;
      s_pi[c & 1][wid] = ci;
    }
    __syncthreads();
    double best = s_pv[c & 1][0];
    int pr = s_pi[c & 1][0];
#pragma unroll
    for (int w = 1; w < SOLVE_WAVES; ++w) {
      const double ov = s_pv[c & 1][w];
      const int oi = s_pi[c & 1][w];
      if (ov > best || (ov == best && oi < pr)) {
        best = ov;
        pr = oi;
      }
    }
    if (best == 0.0) sing = 1;
    if (tid == 0) s_perm[c] = pr;
    cand = -1.0;
    if (row == pr) retired = true;
    if (!retired) {
      const double* Ap = A + pr * lda;
      const double l = Ar[c] / Ap[c];
      // first owned column after c: c + 1 + ((part - (c + 1)) mod parts)
      int d = part - cm;
      if (d < 0) d += parts;
      for (int jb = c + 1 + d; jb < r; jb += SOLVE_BATCH * parts) {
        double pv[SOLVE_BATCH], av[SOLVE_BATCH];
#pragma unroll
        for (int u = 0; u < SOLVE_BATCH; ++u) {
          const int jj = jb + u * parts;
          const bool ok = jj < r;
          pv[u] = ok ? Ap[jj] : 0.0;
          av[u] = ok ? Ar[jj] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < SOLVE_BATCH; ++u) {
          const int jj = jb + u * parts;
          av[u] = fma(-l, pv[u], av[u]);
          if (jj < r) Ar[jj] = av[u];
        }
        if (jb == c + 1) cand = fabs(av[0]);  // this thread owns column c+1 of its row: next pivot candidate
      }
      if (part == 0) b[row] = fma(-l, b[pr], b[row]);
    }
    if (++cm == parts) cm = 0;
  }
  __syncthreads();

  // back substitution over the pivot order: x_c = b[p_c] / A[p_c][c];  b[p_k] -= A[p_k][c] x_c for k < c
  if (wid == 0) {
    // lane owns pivot positions k = lane and lane + 64
    const int p0 = (lane < r) ? s_perm[lane] : 0, p1 = (lane + 64 < r) ? s_perm[lane + 64] : 0;
    double b0 = (lane < r) ? b[p0] : 0.0, b1 = (lane + 64 < r) ? b[p1] : 0.0;
    // reciprocals of the diagonal up front (one division per lane instead of one per column on the serial path)
    const double rd0 = (lane < r) ? 1.0 / A[p0 * lda + lane] : 0.0, rd1 = (lane + 64 < r) ? 1.0 / A[p1 * lda + lane + 64] : 0.0;
    double d0 = A[p0 * lda + r - 1], d1 = A[p1 * lda + r - 1];  // column c of this lane's rows, one column ahead
    for (int c = r - 1; c >= 0; --c) {
      const int src = c & 63;
      const double n0 = (c > 0) ? A[p0 * lda + c - 1] : 0.0, n1 = (c > 0) ? A[p1 * lda + c - 1] : 0.0;
      const double xc = rtw::read_lane(c < 64 ? b0 * rd0 : b1 * rd1, src);
      if (lane < c) b0 = fma(-d0, xc, b0);
      if (lane + 64 < c) b1 = fma(-d1, xc, b1);
      if (lane == src) rb[c] = xc;
      d0 = n0;
      d1 = n1;
    }
  }
  if (info && tid == 0) info[blockIdx.x] = sing ? RT_WARN_SINGULAR : 0;
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
  int parts = SOLVE_THREADS / (int)r;  // threads per row
  if (parts > 8) parts = 8;
  hipLaunchKernelGGL(dense_solve_kernel, dim3((unsigned)B), dim3(SOLVE_THREADS), lds, ctx->stream, K, rhs, (int)r,
                     parts, info);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}
