// Greedy DEIM index selection (src/romtime/deim/deim.py:517-561) as a left-looking,
// partially pivoted LU of the tall collateral basis, entirely on the device.
//
// Reference step k:  c = solve(Phi[p_<k, :k], phi_k[p_<k]);  r = phi_k - Phi[:, :k] c;
//                    p_k = argmax |r|  (first maximum on ties).
// The same residual, written with the residual columns already computed:
//                    r_k = phi_k - sum_{j<k} r_j * (y_j / delta_j),   y = L^-1 phi_k[p_<k],
// with delta_j = r_j[p_j] and L[i][j] = r_j[p_i] / delta_j the unit-lower factor whose
// multipliers are bounded by 1 because every pivot is the residual's largest entry.  L^-1 is
// kept explicitly and bordered step by step, so a step has no sequential k-long dependency
// chain.  Columns are taken in blocks of 8 (see rt_deim_greedy): inside a block a step needs
// only the block's own 8 x 8 corner of L^-1, which every workgroup of the column's chip-wide
// kernel works out for itself before it streams the block's residual columns (16-B coalesced
// loads, column-major) and reduces |r| to (top1, index, top2) per workgroup with a
// lexicographic (value desc, index asc) order -- np.argmax's tie rule.  One launch per step.
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

constexpr int BLK = 8;  // columns per block of the blocked left-looking elimination
constexpr int RES_THREADS = 256;
constexpr int RES_ROWS = 2 * RES_THREADS;  // rows per workgroup of the sweeps (one d2 per thread)
constexpr int COL_ROWS = 4 * RES_THREADS;  // rows per workgroup of a column's kernel (two d2 per thread)
constexpr int PIV_THREADS = 1024;          // the once-per-block serial kernel

struct StepState {   // what a step leaves behind, all in device memory
  int m, nparts;
  // per-workgroup (top1, index, top2) of a residual column, TWO sets: column k's launch writes set k & 1 while all of
  // its workgroups - whenever they are scheduled - still read set (k - 1) & 1 to finish the step before
  double* pv1;
  long* pi1;
  double* pv2;
  long* idx;
  double* delta;
  double* margin;
  double* Linv;  // [m][m] row-major, zero above the diagonal
};

struct StepShared {
  Top2 red[PIV_THREADS / 64];
  double lin[BLK][BLK + 1];   // the block's own corner of L^-1
  double del[BLK];
  long pidx[BLK];
  double l[BLK];
  double b[BLK];
  double yt[BLK];
  long p;
  double v1, v2;
};

// Finish step kp (kp >= jstart, the first column of its block) with T threads, every thread of every workgroup that
// calls it arriving at the same numbers: the argmax over the workgroup partials of residual kp -> p; delta = r_kp[p];
// the block's corner of row kp of L^-1 (l_i = r_i[p] / delta_i for the block's earlier columns i); and, when `k` =
// kp + 1 is in the same block, the coefficients yt_i (i = jstart..kp) of column k against the block's columns -
// column k of R holds t_k, already reduced by the columns before the block.  Everything here is at most BLK x BLK:
// the part of row kp of L^-1 that lies before the block is not needed until the next block starts
// (deim_block_start_kernel).  `persist`: this caller writes idx, delta, margin and the row's corner to memory.
// Nothing read here is written by the launch that calls it: the partials come from the other parity set, t_k from
// the block's t columns `Tb` (written by phase A only), residual columns kp and earlier from finished launches - so
// it does not matter when a workgroup runs relative to the others of its launch (N beyond one round of resident
// workgroups, other streams sharing the chip).
template <int T>
__device__ __forceinline__ void deim_finish_step(const double* R, const double* Tb, long ldr, int kp, int jstart,
                                                 bool want_yt, bool persist, const StepState& a, StepShared& sh) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m = a.m, n = kp - jstart;   // earlier columns of the block
  // corner of L^-1, deltas and indices of the block's earlier columns: independent of the new pivot
  if (tid < BLK * BLK) {
    const int i = tid / BLK, j = tid % BLK;
    sh.lin[i][j] = (i < n && j <= i) ? a.Linv[(long)(jstart + i) * m + jstart + j] : 0.0;
  }
  if (tid < n) {
    sh.del[tid] = a.delta[jstart + tid];
    sh.pidx[tid] = a.idx[jstart + tid];
  }
  Top2 best{-1.0, 0x7fffffffffffffffL, -1.0};
  {
    const long set = (long)(kp & 1) * a.nparts;
    for (int q = tid; q < a.nparts; q += T) best = top2_merge(best, Top2{a.pv1[set + q], a.pi1[set + q], a.pv2[set + q]});
  }
  best = top2_wave(best);
  if (lane == 0) sh.red[wid] = best;
  __syncthreads();
  if (tid == 0) {
    Top2 t = sh.red[0];
    for (int w = 1; w < T / 64; ++w) t = top2_merge(t, sh.red[w]);
    sh.p = t.i1;
    sh.v1 = t.v1;
    sh.v2 = t.v2;
  }
  __syncthreads();
  const long p = sh.p;
  if (tid < n) sh.l[tid] = R[(long)(jstart + tid) * ldr + p] / sh.del[tid];
  if (tid == n) {
    sh.del[n] = R[(long)kp * ldr + p];
    sh.pidx[n] = p;
  }
  if (want_yt && tid >= 64 && tid - 64 <= n) {   // t_k at the block's pivot rows: from the block's t columns, which
    const int j = tid - 64;                       // no column launch writes (column k of R is being overwritten by r_k)
    sh.b[j] = Tb[(long)(kp + 1 - jstart) * ldr + (j < n ? sh.pidx[j] : p)];
  }
  __syncthreads();
  // row kp of the corner:  Linv[kp][j] = -sum_{i=j}^{kp-1} l_i Linv[i][j],  Linv[kp][kp] = 1
  if (tid <= n) {
    double acc = 0.0;
    for (int i = tid; i < n; ++i) acc = fma(sh.l[i], sh.lin[i][tid], acc);
    sh.lin[n][tid] = (tid == n) ? 1.0 : -acc;
  }
  __syncthreads();
  if (want_yt && tid <= n) {   // y_i = sum_{j<=i} Linv[i][j] t_k[p_j],  yt_i = y_i / delta_i
    double acc = 0.0;
    for (int j = 0; j <= tid; ++j) acc = fma(sh.lin[tid][j], sh.b[j], acc);
    sh.yt[tid] = acc / sh.del[tid];
  }
  if (persist) {
    if (tid <= n) a.Linv[(long)kp * m + jstart + tid] = sh.lin[n][tid];
    if (tid == 0) {
      a.idx[kp] = p;
      a.delta[kp] = sh.del[n];
      if (a.margin) a.margin[kp] = (sh.v1 > 0.0) ? (sh.v1 - fmax(sh.v2, 0.0)) / sh.v1 : 0.0;
    }
  }
  __syncthreads();
}

// Column k of a block: first every workgroup finishes step k - 1 for itself (deim_finish_step: the numbers are
// small and all in the L2; doing it here instead of in a single-workgroup kernel between two chip-wide ones takes
// the 9 us "pivot" launch out of every step), then
//   r_k = t_k - sum_{jstart<=j<k} R[j] * yt[j]  (t_k from the block's t columns, r_k into column k of R)  and the
//   per-workgroup top-2 of |r_k|.
__global__ __launch_bounds__(RES_THREADS) void deim_column_kernel(double* __restrict__ R, const double* __restrict__ Tb,
                                                                  long ldr, long N, int k, int jstart, StepState st) {
  __shared__ StepShared sh;
  const int tid = threadIdx.x;
  const int nj = k - jstart;   // < BLK
  constexpr int H = COL_ROWS / RES_ROWS;
  // the column's own loads do not depend on the step before: they are in flight while it is finished
  d2 av[H][BLK - 1], rv[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const long row = (long)blockIdx.x * COL_ROWS + h * RES_ROWS + 2 * tid;
    rv[h] = d2{0.0, 0.0};
#pragma unroll
    for (int j = 0; j < BLK - 1; ++j) av[h][j] = d2{0.0, 0.0};
    if (row < N) {  // ldr is even and columns are padded, so the pair (row, row+1) is always addressable
      const double* col = R + row + (long)jstart * ldr;
#pragma unroll
      for (int j = 0; j < BLK - 1; ++j)
        if (j < nj) av[h][j] = *reinterpret_cast<const d2*>(col + (long)j * ldr);
      rv[h] = *reinterpret_cast<const d2*>(Tb + (long)nj * ldr + row);
    }
  }
  if (nj > 0) deim_finish_step<RES_THREADS>(R, Tb, ldr, k - 1, jstart, true, blockIdx.x == 0, st, sh);
  Top2 best{-1.0, 0x7fffffffffffffffL, -1.0};
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const long row = (long)blockIdx.x * COL_ROWS + h * RES_ROWS + 2 * tid;
    if (row < N) {
      d2 acc{0.0, 0.0};
#pragma unroll
      for (int j = 0; j < BLK - 1; ++j) {
        if (j < nj) {
          const double y = sh.yt[j];
          acc.x = fma(av[h][j].x, y, acc.x);
          acc.y = fma(av[h][j].y, y, acc.y);
        }
      }
      d2 r = rv[h];
      r.x -= acc.x;
      r.y -= acc.y;
      *reinterpret_cast<d2*>(R + (long)k * ldr + row) = r;
      const double ax = fabs(r.x), ay = (row + 1 < N) ? fabs(r.y) : -1.0;
      best = top2_merge(best, top2_merge(Top2{ax, row, -1.0}, Top2{ay, row + 1, -1.0}));
    }
  }
  best = top2_wave(best);
  __syncthreads();   // sh.red is free again
  if ((tid & 63) == 0) sh.red[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) {
    Top2 t = sh.red[0];
    for (int w = 1; w < RES_THREADS / 64; ++w) t = top2_merge(t, sh.red[w]);
    const long slot = (long)(k & 1) * st.nparts + blockIdx.x;
    st.pv1[slot] = t.v1;
    st.pi1[slot] = t.i1;
    st.pv2[slot] = t.v2;
  }
}

// Once per block, one workgroup, k0 = first column of the block that starts (k0 = m: nothing starts):
//   a. finish the last step of the block before (deim_finish_step), whose columns are P = [kb, k0);
//   b. the rows P of L^-1 before column kb, which no step inside P needed:
//        W = L21 Linv[:kb,:kb],  L21[i][q] = r_q[p_i] / delta_q  (i in P, q < kb),
//        Linv[i][j] = -(W[i][j] + sum_{q in P, q < i} l_iq Linv[q][j])   (the step-by-step recurrence, row by row);
//   c. coefficients of the new block's columns against everything before it:
//        YT[c - k0][i] = (Linv[:k0,:k0] b_c)_i / delta_i,   b_c[i] = phi_c[p_i]   (read from the caller's Phi).
__global__ __launch_bounds__(PIV_THREADS) void deim_block_start_kernel(const double* R, long ldr, int k0, int kb, int nb,
                                                                       StepState st, double* __restrict__ YT,
                                                                       const double* __restrict__ Phi, long ld, int layout) {
  constexpr int CG = BLK;  // columns per pass over L^-1: the whole block (one pass; two passes of 4 read every row twice)
  __shared__ StepShared sh;
  __shared__ double s_b[BLK][1024];      // phase b: partial sums of W; phase c: rows [0, CG) = b_c
  __shared__ double s_l21[BLK][1024];
  __shared__ double s_l22[BLK][BLK];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m = st.m, np = k0 - kb;
  deim_finish_step<PIV_THREADS>(R, nullptr, ldr, k0 - 1, kb, false, true, st, sh);
  __threadfence_block();
  __syncthreads();
  if (kb > 0 && k0 < m) {   // (after the last block nobody reads L^-1 again)
    for (int q = tid; q < np * kb; q += PIV_THREADS) {
      const int i = q / kb, c = q % kb;
      s_l21[i][c] = R[(long)c * ldr + st.idx[kb + i]] / st.delta[c];
    }
    if (tid < BLK * BLK) {
      const int i = tid / BLK, q = tid % BLK;   // l_iq for q in P, q < i
      s_l22[i][q] = (i < np && q < i) ? R[(long)(kb + q) * ldr + st.idx[kb + i]] / st.delta[kb + q] : 0.0;
    }
    __syncthreads();
    // W = L21 Linv[:kb,:kb]: eight threads share a column j (rows q == part mod 8 of the walk down L^-1, coalesced
    // across j), their partial sums meet in LDS and are added in a fixed order; then the block's own rows in turn
    constexpr int PARTS = PIV_THREADS / 128;
    double(*s_w)[BLK][128] = reinterpret_cast<double(*)[BLK][128]>(&s_b[0][0]);   // [PARTS][BLK][128] = 64 KB
    const int jj = tid & 127, part = tid >> 7;
    for (int j0 = 0; j0 < kb; j0 += 128) {
      const int j = j0 + jj;
      double w[BLK];
#pragma unroll
      for (int i = 0; i < BLK; ++i) w[i] = 0.0;
      if (j < kb) {
#pragma unroll 4
        for (int q = j + part; q < kb; q += PARTS) {
          const double lq = st.Linv[(long)q * m + j];
#pragma unroll
          for (int i = 0; i < BLK; ++i) w[i] = fma(s_l21[i][q], lq, w[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < BLK; ++i) s_w[part][i][jj] = w[i];
      __syncthreads();
      if (part == 0 && j < kb) {
        double out[BLK];
#pragma unroll
        for (int i = 0; i < BLK; ++i) {
          double acc = 0.0;
#pragma unroll
          for (int q = 0; q < PARTS; ++q) acc += s_w[q][i][jj];
#pragma unroll
          for (int q = 0; q < BLK; ++q)
            if (q < i) acc = fma(s_l22[i][q], out[q], acc);
          out[i] = -acc;
          if (i < np) st.Linv[(long)(kb + i) * m + j] = out[i];
        }
      }
      __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
  }
  if (k0 >= m) return;
  for (int c0 = 0; c0 < nb; c0 += CG) {
    __syncthreads();
    for (int i = tid; i < k0; i += PIV_THREADS) {
      const long p = st.idx[i];
#pragma unroll
      for (int c = 0; c < CG; ++c)   // phi_c at the pivot rows, from the caller's basis
        s_b[c][i] = (c0 + c < nb) ? (layout == RT_COL_MAJOR ? Phi[(long)(k0 + c0 + c) * ld + p] : Phi[p * ld + k0 + c0 + c]) : 0.0;
    }
    __syncthreads();
    // a wave's rows in turn; the first 64 entries of the NEXT row are fetched before this row's sums are reduced (the
    // loads of a row used to start only after the wave-wide reductions of the row before: ~1 us of L2 latency per row)
    double l_next = (wid < k0 && lane <= wid) ? st.Linv[(long)wid * m + lane] : 0.0;
    for (int i = wid; i < k0; i += PIV_THREADS / 64) {
      double acc[CG];
      const double l_first = l_next;
      {
        const int in = i + PIV_THREADS / 64;
        l_next = (in < k0 && lane <= in) ? st.Linv[(long)in * m + lane] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = (lane <= i) ? l_first * s_b[c][lane] : 0.0;
      for (int j = lane + 64; j <= i; j += 64) {
        const double l = st.Linv[(long)i * m + j];
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[c] = fma(l, s_b[c][j], acc[c]);
      }
      const double di = st.delta[i];
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        const double sum = rtw::wave_sum(acc[c]);
        if (lane == 0 && c0 + c < nb) YT[(long)(c0 + c) * m + i] = sum / di;
      }
    }
  }
}

// t_c = phi_c - sum_{j<k0} R_j YT[c][j] for the nb columns of the block in ONE sweep over R[:, :k0]:
// the k0 residual columns are read once per block instead of once per column.  The t columns go to the block's own
// buffer Tb (column-major, ldr): the column launches read them there and write the residuals into R, so no launch
// reads what it writes.  k0 = 0: a copy of the first block's columns.
__global__ __launch_bounds__(RES_THREADS) void deim_phase_a_kernel(const double* __restrict__ R, double* __restrict__ Tb,
                                                                   long ldr, long N, int k0, int nb, int m,
                                                                   const double* __restrict__ YT,
                                                                   const double* __restrict__ Phi, long ld, int layout) {
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
  // phi_c comes straight from the caller's basis, in its own layout (row-major: the block's 8 columns are 64 contiguous
  // bytes of a row) - R is never initialised with a transposed copy of Phi (that copy was 0.19 of C4's 2.5 ms)
  const bool two = row + 1 < N;
  double p0[BLK], p1[BLK];
  if (layout == RT_COL_MAJOR) {
#pragma unroll
    for (int c = 0; c < BLK; ++c) {
      const double* src = Phi + (long)(k0 + (c < nb ? c : 0)) * ld + row;
      p0[c] = src[0];
      p1[c] = two ? src[1] : 0.0;
    }
  } else {
    const double* s0 = Phi + row * ld + k0;
    const double* s1 = s0 + (two ? ld : 0);
    if (nb == BLK && ((ld | k0) & 1) == 0 && (reinterpret_cast<size_t>(Phi) & 15) == 0) {
#pragma unroll
      for (int c = 0; c < BLK; c += 2) {
        const d2 a = *reinterpret_cast<const d2*>(s0 + c), b = *reinterpret_cast<const d2*>(s1 + c);
        p0[c] = a.x; p0[c + 1] = a.y;
        p1[c] = two ? b.x : 0.0; p1[c + 1] = two ? b.y : 0.0;
      }
    } else {
#pragma unroll
      for (int c = 0; c < BLK; ++c) {
        const int cc = c < nb ? c : 0;
        p0[c] = s0[cc];
        p1[c] = two ? s1[cc] : 0.0;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < BLK; ++c) {
    if (c < nb) {
      d2 r{p0[c] - acc[c].x, p1[c] - acc[c].y};
      *reinterpret_cast<d2*>(Tb + (long)c * ldr + row) = r;
    }
  }
}

// PT_U[i][j] = Phi[p_i][j]
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
  const int nparts = (int)((N + RES_ROWS - 1) / RES_ROWS);   // workgroups of the block sweeps
  const int ncol = (int)((N + COL_ROWS - 1) / COL_ROWS);     // workgroups (= argmax partials) of a column's kernel
  // scratch: R | Tb | Linv | delta | YT | pv1 | pv2 | pi1  (the three partial arrays twice: one set per column parity)
  size_t off = 0;
  auto take = [&off](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t oR = take(sizeof(double) * ldr * m), oT = take(sizeof(double) * ldr * BLK),
               oL = take(sizeof(double) * m * m), oD = take(sizeof(double) * m), oYT = take(sizeof(double) * BLK * m),
               oV1 = take(sizeof(double) * 2 * ncol), oV2 = take(sizeof(double) * 2 * ncol),
               oI1 = take(sizeof(long) * 2 * ncol);
  void* base = nullptr;
  int rc = rt_scratch2(ctx, off, &base);
  if (rc != RT_OK) return rc;
  char* b8 = static_cast<char*>(base);
  double* R = reinterpret_cast<double*>(b8 + oR);
  double* Tb = reinterpret_cast<double*>(b8 + oT);
  double* Linv = reinterpret_cast<double*>(b8 + oL);
  double* delta = reinterpret_cast<double*>(b8 + oD);
  double* YT = reinterpret_cast<double*>(b8 + oYT);
  hipStream_t st = ctx->stream;

  RT_HIP_CHECK(ctx, hipMemsetAsync(Linv, 0, sizeof(double) * m * m, st));
  // Blocked left-looking elimination: per block of BLK columns one serial kernel (finish the block before, block
  // coefficients) and one sweep (phase A) that applies all earlier residual columns, then one launch per column:
  // the column's kernel finishes the step before it in every workgroup and reduces the column against at most
  // BLK-1 in-block columns.  HBM traffic drops from ~4 N m^2 to ~4 N m^2 / BLK + 4 N m (BLK + 7) bytes.
  const StepState state{(int)m, ncol, reinterpret_cast<double*>(b8 + oV1), reinterpret_cast<long*>(b8 + oI1),
                        reinterpret_cast<double*>(b8 + oV2), reinterpret_cast<long*>(idx), delta, margin, Linv};
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&deim_phase_a_kernel), 1024 * BLK * 8));
  for (int k0 = 0; k0 < (int)m; k0 += BLK) {
    const int nb = ((int)m - k0 < BLK) ? (int)m - k0 : BLK;
    if (k0 > 0)
      hipLaunchKernelGGL(deim_block_start_kernel, dim3(1), dim3(PIV_THREADS), 0, st, R, ldr, k0, k0 - BLK, nb, state, YT, Phi,
                         (long)ld, layout);
    hipLaunchKernelGGL(deim_phase_a_kernel, dim3(nparts), dim3(RES_THREADS), sizeof(double) * k0 * BLK, st, R, Tb, ldr,
                       (long)N, k0, nb, (int)m, YT, Phi, (long)ld, layout);
    for (int k = k0; k < k0 + nb; ++k)
      hipLaunchKernelGGL(deim_column_kernel, dim3(ncol), dim3(RES_THREADS), 0, st, R, Tb, ldr, (long)N, k, k0, state);
  }
  {  // the last step
    const int kb = ((int)m - 1) / BLK * BLK;
    hipLaunchKernelGGL(deim_block_start_kernel, dim3(1), dim3(PIV_THREADS), 0, st, R, ldr, (int)m, kb, 0, state, YT, Phi,
                       (long)ld, layout);
  }
  RT_HIP_CHECK(ctx, hipGetLastError());
  if (PT_U) {
    hipLaunchKernelGGL(deim_gather_ptu_kernel, dim3((unsigned)m), dim3(128), 0, st, Phi, (long)ld, layout, (int)m,
                       reinterpret_cast<const long*>(idx), PT_U);
    RT_HIP_CHECK(ctx, hipGetLastError());
  }
  RT_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return RT_OK;
}
