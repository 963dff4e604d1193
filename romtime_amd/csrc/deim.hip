// Greedy DEIM index selection (src/romtime/deim/deim.py:517-561) as a left-looking,
// partially pivoted LU of the tall collateral basis, entirely on the device.
//
// Reference step k:  c = solve(Phi[p_<k, :k], phi_k[p_<k]);  r = phi_k - Phi[:, :k] c;
//                    p_k = argmax |r|  (first maximum on ties).
// The same residual, written with the residual columns already computed:
//                    r_k = phi_k - sum_{j<k} r_j * (y_j / delta_j),   y = L^-1 phi_k[p_<k],
// with delta_j = r_j[p_j] and L[i][j] = r_j[p_i] / delta_j the unit-lower factor whose
// multipliers are bounded by 1 because every pivot is the residual's largest entry.  L^-1 is
// kept explicitly and bordered by one vector-matrix product per step, so a step has no
// sequential k-long dependency chain: one single-workgroup "pivot" kernel (argmax finish,
// border, triangular mat-vec) and one chip-wide "residual" kernel that streams k residual
// columns (16-B coalesced loads, column-major) and reduces |r| to (top1, index, top2) per
// workgroup with a lexicographic (value desc, index asc) order -- np.argmax's tie rule.
// HBM traffic is the algorithmic 8 N (k+2) bytes per step.
#include "common.h"
#include "wave_ops.h"

typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

struct Top2 {
  double v1;
  long i1;
  double v2;
};

__device__ __forceinline__ Top2 top2_merge(const Top2& a, const Top2& b) {
  Top2 o;
  const bool a_first = (a.v1 > b.v1) || (a.v1 == b.v1 && a.i1 <= b.i1);
  if (a_first) {
    o.v1 = a.v1;
    o.i1 = a.i1;
    o.v2 = fmax(a.v2, b.v1);
  } else {
    o.v1 = b.v1;
    o.i1 = b.i1;
    o.v2 = fmax(b.v2, a.v1);
  }
  return o;
}

// Wave-wide merge on the DPP path (wave_ops.h): lanes without a source receive the identity {-1, max, -1}.
// The merge is associative and commutative on (value desc, index asc), so the result is independent of the tree.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ Top2 top2_dpp(const Top2& t) {
  const long long v1 = __double_as_longlong(t.v1), v2 = __double_as_longlong(t.v2);
  constexpr int NEG1_HI = (int)0xBFF00000;  // -1.0
  Top2 o;
  const int v1lo = __builtin_amdgcn_update_dpp(0, (int)v1, CTRL, ROW_MASK, 0xf, false);
  const int v1hi = __builtin_amdgcn_update_dpp(NEG1_HI, (int)(v1 >> 32), CTRL, ROW_MASK, 0xf, false);
  const int v2lo = __builtin_amdgcn_update_dpp(0, (int)v2, CTRL, ROW_MASK, 0xf, false);
  const int v2hi = __builtin_amdgcn_update_dpp(NEG1_HI, (int)(v2 >> 32), CTRL, ROW_MASK, 0xf, false);
  const int ilo = __builtin_amdgcn_update_dpp(-1, (int)t.i1, CTRL, ROW_MASK, 0xf, false);
  const int ihi = __builtin_amdgcn_update_dpp(0x7fffffff, (int)(t.i1 >> 32), CTRL, ROW_MASK, 0xf, false);
  o.v1 = __longlong_as_double(((long long)v1hi << 32) | (unsigned)v1lo);
  o.v2 = __longlong_as_double(((long long)v2hi << 32) | (unsigned)v2lo);
  o.i1 = ((long)ihi << 32) | (unsigned)ilo;
  return o;
}

__device__ __forceinline__ Top2 top2_wave(Top2 t) {  // result valid in lane 63, returned wave-uniform
  t = top2_merge(t, top2_dpp<0x111>(t));        // row_shr:1
  t = top2_merge(t, top2_dpp<0x112>(t));        // row_shr:2
  t = top2_merge(t, top2_dpp<0x114>(t));        // row_shr:4
  t = top2_merge(t, top2_dpp<0x118>(t));        // row_shr:8  -> lane 15 of each row: the row's result
  t = top2_merge(t, top2_dpp<0x142, 0xa>(t));   // row_bcast:15
  t = top2_merge(t, top2_dpp<0x143, 0xc>(t));   // row_bcast:31 -> lane 63
  Top2 o;
  o.v1 = rtw::read_lane(t.v1, 63);
  o.v2 = rtw::read_lane(t.v2, 63);
  const int ilo = __builtin_amdgcn_readlane((int)t.i1, 63), ihi = __builtin_amdgcn_readlane((int)(t.i1 >> 32), 63);
  o.i1 = ((long)ihi << 32) | (unsigned)ilo;
  return o;
}

constexpr int RES_THREADS = 256;
constexpr int RES_ROWS = 2 * RES_THREADS;  // rows per workgroup (one d2 per thread)

// r_k = R[k] - sum_{j<k} R[j] * yt[j]  (in place in column k);  per-workgroup top-2 of |r_k|.
__global__ __launch_bounds__(RES_THREADS) void deim_residual_kernel(double* __restrict__ R, long ldr, long N,
                                                                    int k, int jstart,
                                                                    const double* __restrict__ yt,
                                                                    double* __restrict__ pv1,
                                                                    long* __restrict__ pi1,
                                                                    double* __restrict__ pv2) {
  __shared__ double s_yt[1024];
  __shared__ Top2 s_red[RES_THREADS / 64];
  const int tid = threadIdx.x;
  for (int j = jstart + tid; j < k; j += RES_THREADS) s_yt[j] = yt[j];
  __syncthreads();
  const long row = (long)blockIdx.x * RES_ROWS + 2 * tid;
  Top2 best{-1.0, 0x7fffffffffffffffL, -1.0};
  if (row < N) {  // ldr is even and columns are padded, so the pair (row, row+1) is always addressable
    d2 acc{0.0, 0.0};
    const double* col = R + row;
    int j = jstart;
    for (; j + 4 <= k; j += 4) {
      const d2 a0 = *reinterpret_cast<const d2*>(col + (long)(j + 0) * ldr);
      const d2 a1 = *reinterpret_cast<const d2*>(col + (long)(j + 1) * ldr);
      const d2 a2 = *reinterpret_cast<const d2*>(col + (long)(j + 2) * ldr);
      const d2 a3 = *reinterpret_cast<const d2*>(col + (long)(j + 3) * ldr);
      const double y0 = s_yt[j], y1 = s_yt[j + 1], y2 = s_yt[j + 2], y3 = s_yt[j + 3];
      acc.x = fma(a0.x, y0, acc.x); acc.y = fma(a0.y, y0, acc.y);
      acc.x = fma(a1.x, y1, acc.x); acc.y = fma(a1.y, y1, acc.y);
      acc.x = fma(a2.x, y2, acc.x); acc.y = fma(a2.y, y2, acc.y);
      acc.x = fma(a3.x, y3, acc.x); acc.y = fma(a3.y, y3, acc.y);
    }
    for (; j < k; ++j) {
      const d2 a = *reinterpret_cast<const d2*>(col + (long)j * ldr);
      const double y = s_yt[j];
      acc.x = fma(a.x, y, acc.x);
      acc.y = fma(a.y, y, acc.y);
    }
    d2* dst = reinterpret_cast<d2*>(R + (long)k * ldr + row);
    d2 r = *dst;
    r.x -= acc.x;
    r.y -= acc.y;
    *dst = r;
    const double ax = fabs(r.x), ay = (row + 1 < N) ? fabs(r.y) : -1.0;
    Top2 t0{ax, row, -1.0}, t1{ay, row + 1, -1.0};
    best = top2_merge(t0, t1);
  }
  best = top2_wave(best);
  if ((tid & 63) == 0) s_red[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) {
    Top2 t = s_red[0];
    for (int w = 1; w < RES_THREADS / 64; ++w) t = top2_merge(t, s_red[w]);
    pv1[blockIdx.x] = t.v1;
    pi1[blockIdx.x] = t.i1;
    pv2[blockIdx.x] = t.v2;
  }
}

constexpr int PIV_THREADS = 1024;

// Runs between residual kernels.  `k` = the step whose coefficients are prepared (k >= 1);
// first finishes step k-1: idx[k-1], delta[k-1], margin[k-1], row k-1 of L^-1.
// With k == m only the finish part runs.
__global__ __launch_bounds__(PIV_THREADS) void deim_pivot_kernel(const double* __restrict__ R, long ldr, int k,
                                                                 int jstart, int m, int nparts, const double* __restrict__ pv1,
                                                                 const long* __restrict__ pi1,
                                                                 const double* __restrict__ pv2, long* idx,
                                                                 double* delta, double* margin, double* Linv,
                                                                 double* yt) {
  __shared__ Top2 s_red[PIV_THREADS / 64];
  __shared__ double s_l[1024];
  __shared__ double s_b[1024];
  __shared__ double s_part[PIV_THREADS];
  __shared__ long s_p;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kp = k - 1;  // step being finished

  // 1. final argmax over the workgroup partials of residual kp
  Top2 best{-1.0, 0x7fffffffffffffffL, -1.0};
  for (int q = tid; q < nparts; q += PIV_THREADS) best = top2_merge(best, Top2{pv1[q], pi1[q], pv2[q]});
  best = top2_wave(best);
  if (lane == 0) s_red[wid] = best;
  __syncthreads();
  if (tid == 0) {
    Top2 t = s_red[0];
    for (int w = 1; w < PIV_THREADS / 64; ++w) t = top2_merge(t, s_red[w]);
    s_p = t.i1;
    idx[kp] = t.i1;
    delta[kp] = R[(long)kp * ldr + t.i1];
    if (margin) margin[kp] = (t.v1 > 0.0) ? (t.v1 - fmax(t.v2, 0.0)) / t.v1 : 0.0;
  }
  __syncthreads();
  const long p = s_p;

  // 2. border L^-1 with row kp:  l_j = r_j[p] / delta_j (j < kp);  Linv[kp][:] = -l^T Linv, Linv[kp][kp] = 1.
  //    Eight threads share a column (rows i == part mod 8), so nobody walks more than kp/8 entries of the
  //    column-strided Linv; the partial sums meet in LDS and are added in a fixed order.
  for (int j = tid; j < kp; j += PIV_THREADS) s_l[j] = R[(long)j * ldr + p] / delta[j];
  __syncthreads();
  {
    const int jj = tid & 127, part = tid >> 7;
    for (int j0 = 0; j0 < kp; j0 += 128) {
      const int j = j0 + jj;
      double acc = 0.0;
      if (j < kp) {
#pragma unroll 4
        for (int i = j + part; i < kp; i += PIV_THREADS / 128) acc = fma(s_l[i], Linv[(long)i * m + j], acc);
      }
      s_part[part * 128 + jj] = acc;
      __syncthreads();
      if (part == 0 && j < kp) {
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < PIV_THREADS / 128; ++q) sum += s_part[q * 128 + jj];
        Linv[(long)kp * m + j] = -sum;
      }
      __syncthreads();
    }
  }
  if (tid == 0) Linv[(long)kp * m + kp] = 1.0;
  if (k >= m) return;
  __threadfence_block();
  __syncthreads();

  // 3. in-block coefficients: column k of R holds t_k (already reduced by the columns before the block,
  //    see deim_phase_a_kernel), so  y_i = sum_{jstart<=j<=i} Linv[i][j] t_k[p_j]  and  yt_i = y_i / delta_i
  for (int i = jstart + tid; i < k; i += PIV_THREADS) s_b[i] = R[(long)k * ldr + idx[i]];
  __syncthreads();
  for (int i = jstart + wid; i < k; i += PIV_THREADS / 64) {
    double acc = 0.0;
    for (int j = jstart + lane; j <= i; j += 64) acc = fma(Linv[(long)i * m + j], s_b[j], acc);
    acc = rtw::wave_sum(acc);
    if (lane == 0) yt[i] = acc / delta[i];
  }
}

constexpr int BLK = 8;  // columns per block of the blocked left-looking elimination

// Coefficients of a block of columns against everything before the block:
//   YT[c - k0][i] = (Linv[:k0,:k0] b_c)_i / delta_i,   b_c[i] = phi_c[p_i]   (i < k0, c in the block)
// Column c of R still holds phi_c when this runs.
__global__ __launch_bounds__(PIV_THREADS) void deim_block_coeff_kernel(const double* __restrict__ R, long ldr, int k0,
                                                                       int nb, int m, const long* __restrict__ idx,
                                                                       const double* __restrict__ delta,
                                                                       const double* __restrict__ Linv,
                                                                       double* __restrict__ YT) {
  constexpr int CG = 4;  // columns per pass over L^-1 (each row of L^-1 is read once per pass)
  __shared__ double s_b[CG][1024];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int c0 = 0; c0 < nb; c0 += CG) {
    __syncthreads();
    for (int i = tid; i < k0; i += PIV_THREADS) {
      const long p = idx[i];
#pragma unroll
      for (int c = 0; c < CG; ++c) s_b[c][i] = (c0 + c < nb) ? R[(long)(k0 + c0 + c) * ldr + p] : 0.0;
    }
    __syncthreads();
    for (int i = wid; i < k0; i += PIV_THREADS / 64) {
      double acc[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = 0.0;
      for (int j = lane; j <= i; j += 64) {
        const double l = Linv[(long)i * m + j];
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[c] = fma(l, s_b[c][j], acc[c]);
      }
      const double di = delta[i];
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        const double sum = rtw::wave_sum(acc[c]);
        if (lane == 0 && c0 + c < nb) YT[(long)(c0 + c) * m + i] = sum / di;
      }
    }
  }
}

// t_c = phi_c - sum_{j<k0} R_j YT[c][j] for the nb columns of the block in ONE sweep over R[:, :k0]:
// the k0 residual columns are read once per block instead of once per column.
__global__ __launch_bounds__(RES_THREADS) void deim_phase_a_kernel(double* __restrict__ R, long ldr, long N, int k0,
                                                                   int nb, int m, const double* __restrict__ YT) {
  extern __shared__ double s_yt[];  // [k0][BLK]
  const int tid = threadIdx.x;
  for (int q = tid; q < k0 * BLK; q += RES_THREADS) {
    const int j = q / BLK, c = q % BLK;
    s_yt[q] = (c < nb) ? YT[(long)c * m + j] : 0.0;
  }
  __syncthreads();
  const long row = (long)blockIdx.x * RES_ROWS + 2 * tid;
  if (row >= N) return;
  d2 acc[BLK];
#pragma unroll
  for (int c = 0; c < BLK; ++c) acc[c] = d2{0.0, 0.0};
  const double* col = R + row;
#pragma unroll 2
  for (int j = 0; j < k0; ++j) {
    const d2 a = *reinterpret_cast<const d2*>(col + (long)j * ldr);
    const double* y = s_yt + j * BLK;
#pragma unroll
    for (int c = 0; c < BLK; ++c) {
      acc[c].x = fma(a.x, y[c], acc[c].x);
      acc[c].y = fma(a.y, y[c], acc[c].y);
    }
  }
#pragma unroll
  for (int c = 0; c < BLK; ++c) {
    if (c < nb) {
      d2* dst = reinterpret_cast<d2*>(R + (long)(k0 + c) * ldr + row);
      d2 r = *dst;
      r.x -= acc[c].x;
      r.y -= acc[c].y;
      *dst = r;
    }
  }
}

// R (column-major, ldr) <- Phi in either layout
__global__ void deim_copy_in_kernel(const double* __restrict__ Phi, long ld, int layout, long N, int m,
                                    double* __restrict__ R, long ldr) {
  __shared__ double tile[32][33];
  if (layout == RT_COL_MAJOR) {
    const long i = (long)blockIdx.x * blockDim.x * blockDim.y + threadIdx.y * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i < ldr) R[(long)j * ldr + i] = (i < N) ? Phi[(long)j * ld + i] : 0.0;
    return;
  }
  // row-major: 32x32 tile transpose; blockIdx.x -> row tile, blockIdx.y -> column tile
  const long i0 = (long)blockIdx.x * 32;
  const int j0 = blockIdx.y * 32;
  for (int rr = threadIdx.y; rr < 32; rr += blockDim.y) {
    const long i = i0 + rr;
    const int j = j0 + threadIdx.x;
    tile[rr][threadIdx.x] = (i < N && j < m) ? Phi[i * ld + j] : 0.0;
  }
  __syncthreads();
  for (int cc = threadIdx.y; cc < 32; cc += blockDim.y) {
    const int j = j0 + cc;
    const long i = i0 + threadIdx.x;
    if (j < m && i < ldr) R[(long)j * ldr + i] = tile[threadIdx.x][cc];
  }
}

__global__ void deim_gather_ptu_kernel(const double* __restrict__ Phi, long ld, int layout, int m,
                                       const long* __restrict__ idx, double* __restrict__ PT_U) {
  const int i = blockIdx.x;
  const long row = idx[i];
  for (int j = threadIdx.x; j < m; j += blockDim.x)
    PT_U[(long)i * m + j] = (layout == RT_COL_MAJOR) ? Phi[(long)j * ld + row] : Phi[row * ld + j];
}

}  // namespace

extern "C" int rt_deim_greedy(rt_ctx* ctx, const double* Phi, int64_t N, int64_t m, int64_t ld, int layout,
                              int64_t* idx, double* PT_U, double* margin) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, Phi && idx && N >= 1 && m >= 1);
  RT_ARG_CHECK(ctx, m <= 1024 && m <= N);
  RT_ARG_CHECK(ctx, layout == RT_ROW_MAJOR || layout == RT_COL_MAJOR);
  RT_ARG_CHECK(ctx, ld >= (layout == RT_COL_MAJOR ? N : m));

  const long ldr = (N + 15) / 16 * 16;
  const int nparts = (int)((N + RES_ROWS - 1) / RES_ROWS);
  // scratch: R | Linv | yt | delta | pv1 | pv2 | pi1
  size_t off = 0;
  auto take = [&off](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t oR = take(sizeof(double) * ldr * m), oL = take(sizeof(double) * m * m), oY = take(sizeof(double) * m),
               oD = take(sizeof(double) * m), oYT = take(sizeof(double) * BLK * m), oV1 = take(sizeof(double) * nparts),
               oV2 = take(sizeof(double) * nparts), oI1 = take(sizeof(long) * nparts);
  void* base = nullptr;
  int rc = rt_scratch2(ctx, off, &base);
  if (rc != RT_OK) return rc;
  char* b8 = static_cast<char*>(base);
  double* R = reinterpret_cast<double*>(b8 + oR);
  double* Linv = reinterpret_cast<double*>(b8 + oL);
  double* yt = reinterpret_cast<double*>(b8 + oY);
  double* delta = reinterpret_cast<double*>(b8 + oD);
  double* YT = reinterpret_cast<double*>(b8 + oYT);
  double* pv1 = reinterpret_cast<double*>(b8 + oV1);
  double* pv2 = reinterpret_cast<double*>(b8 + oV2);
  long* pi1 = reinterpret_cast<long*>(b8 + oI1);
  hipStream_t st = ctx->stream;

  RT_HIP_CHECK(ctx, hipMemsetAsync(Linv, 0, sizeof(double) * m * m, st));
  if (layout == RT_COL_MAJOR) {
    dim3 grid((unsigned)((ldr + 255) / 256), (unsigned)m);
    hipLaunchKernelGGL(deim_copy_in_kernel, grid, dim3(64, 4), 0, st, Phi, (long)ld, layout, (long)N, (int)m, R, ldr);
  } else {
    dim3 grid((unsigned)((ldr + 31) / 32), (unsigned)((m + 31) / 32));
    hipLaunchKernelGGL(deim_copy_in_kernel, grid, dim3(32, 8), 0, st, Phi, (long)ld, layout, (long)N, (int)m, R, ldr);
  }
  RT_HIP_CHECK(ctx, hipGetLastError());

  // Blocked left-looking elimination: per block of BLK columns one sweep (phase A) applies all earlier
  // residual columns, then the columns of the block are finished one by one against at most BLK-1
  // in-block columns.  HBM traffic drops from ~4 N m^2 to ~4 N m^2 / BLK + 4 N m (BLK + 7) bytes.
  long* idxp = reinterpret_cast<long*>(idx);
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&deim_phase_a_kernel), 1024 * BLK * 8));
  for (int k0 = 0; k0 < (int)m; k0 += BLK) {
    const int nb = ((int)m - k0 < BLK) ? (int)m - k0 : BLK;
    if (k0 > 0) {
      hipLaunchKernelGGL(deim_pivot_kernel, dim3(1), dim3(PIV_THREADS), 0, st, R, ldr, k0, k0, (int)m, nparts, pv1,
                         pi1, pv2, idxp, delta, margin, Linv, yt);
      hipLaunchKernelGGL(deim_block_coeff_kernel, dim3(1), dim3(PIV_THREADS), 0, st, R, ldr, k0, nb, (int)m, idxp,
                         delta, Linv, YT);
      hipLaunchKernelGGL(deim_phase_a_kernel, dim3(nparts), dim3(RES_THREADS), sizeof(double) * k0 * BLK, st, R, ldr,
                         (long)N, k0, nb, (int)m, YT);
    }
    for (int k = k0; k < k0 + nb; ++k) {
      if (k > k0)
        hipLaunchKernelGGL(deim_pivot_kernel, dim3(1), dim3(PIV_THREADS), 0, st, R, ldr, k, k0, (int)m, nparts, pv1,
                           pi1, pv2, idxp, delta, margin, Linv, yt);
      hipLaunchKernelGGL(deim_residual_kernel, dim3(nparts), dim3(RES_THREADS), 0, st, R, ldr, (long)N, k, k0, yt,
                         pv1, pi1, pv2);
    }
  }
  hipLaunchKernelGGL(deim_pivot_kernel, dim3(1), dim3(PIV_THREADS), 0, st, R, ldr, (int)m, (int)m, (int)m, nparts,
                     pv1, pi1, pv2, idxp, delta, margin, Linv, yt);
  RT_HIP_CHECK(ctx, hipGetLastError());
  if (PT_U) {
    hipLaunchKernelGGL(deim_gather_ptu_kernel, dim3((unsigned)m), dim3(128), 0, st, Phi, (long)ld, layout, (int)m,
                       reinterpret_cast<const long*>(idx), PT_U);
    RT_HIP_CHECK(ctx, hipGetLastError());
  }
  RT_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return RT_OK;
}
