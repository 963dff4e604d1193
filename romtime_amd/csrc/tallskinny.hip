// Tall-skinny product  Y (N x k) = X (N x n, row-major) T (n x k),  k <= 128: the POD back-projection
// U_r = X W (pod.py:38 folded into the Gram route, SURVEY 8d "POD pass 2") and the lift u_h = V u_N.
//
// 8 N (n + k) bytes for 2 N n k flops: at k = 40 (three 16-wide MFMA column tiles) the matrix-core time is
// within 15 % of the HBM time, so the kernel is built to overlap the two as well as possible rather than for
// either alone: a workgroup takes 64 rows and walks the contraction in stages of 32 columns (256 contiguous
// bytes per row), X and T stages go HBM -> registers -> LDS (next stage in flight while the current one is
// multiplied), 4 waves x 16 rows, each wave all ceil(k/16) column tiles; 29 KB of LDS and <= 128 VGPRs leave room
// for four workgroups per CU.  Measured 4.1 TB/s at 1e6 x 512 -> 40 (generic skinny tile: 3.8; 64-column stages
// with two workgroups per CU: 3.8).
#include <cstdlib>

#include "common.h"

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TS_THREADS = 256;
constexpr int TS_KS = 32;      // contraction columns per stage
constexpr int TS_SA = 34;      // LDS stride of the X stage ([row][k]): 2 SA == 4 (mod 8) -> the 16 rows of an
                               // A-operand read fall in distinct banks

struct TsParams {
  const double* X;
  const double* T;
  double* Y;
  long N, ldx, ldt, ldy;
  int n, k;
  int no_fast;   // ROMTIME_TS_FLAGS & 1: measurement switch, general refill path only
  int nt_x, nt_y;  // ROMTIME_TS_FLAGS & 8 / & 16: non-temporal loads of X / stores of Y (measurement switches)
};

// RB = 16-row blocks per wave: a workgroup takes 64 RB rows.  With RB = 2 a T stage (re-read from L2 by every workgroup)
// serves twice as many rows: L2 -> CU traffic per row of X drops from 1.75x to 1.37x of the X bytes at k = 40.
template <int NT, int RB>
__global__ __launch_bounds__(TS_THREADS, RB == 2 ? (NT <= 4 ? 3 : 1) : (NT <= 4 ? 4 : 2)) void tallskinny_kernel(const TsParams p) {
  constexpr int TS_BM = 64 * RB;
  constexpr int TS_XL = TS_BM * TS_KS / 2 / TS_THREADS;      // d2 loads of X per thread and stage
  constexpr int KP = 16 * NT;                                 // padded output width
  constexpr int TL = (TS_KS * KP / 2 + TS_THREADS - 1) / TS_THREADS;  // d2 loads of T per thread and stage
  __shared__ __attribute__((aligned(16))) double sA[TS_BM * TS_SA];
  __shared__ __attribute__((aligned(16))) double sT[TS_KS * KP];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long row0 = (long)blockIdx.x * TS_BM;
  const bool xvec = ((p.ldx & 1) == 0) && ((reinterpret_cast<size_t>(p.X) & 15) == 0);
  const bool tvec = ((p.ldt & 1) == 0) && ((reinterpret_cast<size_t>(p.T) & 15) == 0);

  d4 acc[RB][NT];
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[b][j] = d4{0.0, 0.0, 0.0, 0.0};

  d2 xr[TS_XL], tr[TL];
  auto fetch = [&](int c0) {  // stage of contraction columns c0 .. c0 + TS_KS - 1 into registers
#pragma unroll
    for (int i = 0; i < TS_XL; ++i) {
      const int q = tid + TS_THREADS * i, r = q / (TS_KS / 2), c = c0 + 2 * (q % (TS_KS / 2));
      const long row = row0 + r;
      d2 v{0.0, 0.0};
      if (row < p.N) {
        const double* src = p.X + row * p.ldx + c;
        if (xvec && c + 1 < p.n) {
          v = *reinterpret_cast<const d2*>(src);
        } else {
          if (c < p.n) v.x = src[0];
          if (c + 1 < p.n) v.y = src[1];
        }
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < TL; ++i) {
      const int q = tid + TS_THREADS * i, kk = q / (KP / 2), j = 2 * (q % (KP / 2));
      d2 v{0.0, 0.0};
      if (kk < TS_KS && c0 + kk < p.n) {
        const double* src = p.T + (long)(c0 + kk) * p.ldt + j;
        if (tvec && j + 1 < p.k) {
          v = *reinterpret_cast<const d2*>(src);
        } else {
          if (j < p.k) v.x = src[0];
          if (j + 1 < p.k) v.y = src[1];
        }
      }
      tr[i] = v;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < TS_XL; ++i) {
      const int q = tid + TS_THREADS * i, r = q / (TS_KS / 2), c = 2 * (q % (TS_KS / 2));
      *reinterpret_cast<d2*>(&sA[r * TS_SA + c]) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < TL; ++i) {
      const int q = tid + TS_THREADS * i, kk = q / (KP / 2), j = 2 * (q % (KP / 2));
      if (kk < TS_KS) *reinterpret_cast<d2*>(&sT[kk * KP + j]) = tr[i];
    }
  };

  fetch(0);
  commit();
  __syncthreads();
  const double* fa = sA + (16 * wid + l15) * TS_SA + l4;  // A operand: row 16 (w + 4 b) + l15, k = 4 k4 + l4
  const double* fb = sT + l4 * KP + l15;                   // B operand: k = 4 k4 + l4, column 16 j + l15
  int c0 = 0;
  // Fast loop for workgroups whose 64 rows exist and whose stages are whole (n a multiple of 32, 16-byte aligned pairs):
  // the refill addresses are a wave-uniform base, advanced by scalar adds, plus per-thread byte offsets computed once
  // - the general `fetch` spends ~10 VALU instructions per load on 64-bit address arithmetic and predicates, and an FP64
  // MFMA cannot overlap with VALU work of its SIMD (24 MFMAs per wave and stage here: the kernel was VALU-bound).
  const bool fast = !p.no_fast && xvec && tvec && (row0 + TS_BM <= p.N) && (p.n % TS_KS == 0) && ((p.k & 1) == 0) &&
                    ((long)TS_BM * p.ldx * 8 < (1L << 31)) && ((long)TS_KS * p.ldt * 8 < (1L << 31));
  if (fast) {
    const char* gx = reinterpret_cast<const char*>(p.X + row0 * p.ldx + TS_KS);   // stage 1 of this workgroup's rows
    const char* gt = reinterpret_cast<const char*>(p.T + (long)TS_KS * p.ldt);
    const unsigned xoff = (unsigned)(((long)(tid / (TS_KS / 2)) * p.ldx + 2 * (tid % (TS_KS / 2))) * 8);
    const long xstep = (long)(TS_THREADS / (TS_KS / 2)) * p.ldx * 8;               // load i: 16 i rows further down
    unsigned toff[TL];
    bool tuse[TL];
#pragma unroll
    for (int i = 0; i < TL; ++i) {
      const int q = tid + TS_THREADS * i, kk = q / (KP / 2), j = 2 * (q % (KP / 2));
      toff[i] = (unsigned)(((long)kk * p.ldt + j) * 8);
      tuse[i] = j + 1 < p.k;                                                        // padded columns stay zero
      tr[i] = d2{0.0, 0.0};
    }
    const long tstage = (long)TS_KS * p.ldt * 8;
    for (; c0 + TS_KS < p.n; c0 += TS_KS) {
#pragma unroll
      for (int i = 0; i < TS_XL; ++i) {
        const d2* src = reinterpret_cast<const d2*>(gx + i * xstep + xoff);
        xr[i] = p.nt_x ? __builtin_nontemporal_load(src) : *src;
      }
#pragma unroll
      for (int i = 0; i < TL; ++i)
        if (tuse[i]) tr[i] = *reinterpret_cast<const d2*>(gt + toff[i]);
#pragma unroll
      for (int k4 = 0; k4 < TS_KS / 4; ++k4) {
        double bq[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bq[j] = fb[4 * k4 * KP + 16 * j];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          const double a = fa[64 * b * TS_SA + 4 * k4];
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[b][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[j], acc[b][j], 0, 0, 0);
        }
      }
      gx += TS_KS * 8;
      gt += tstage;
      __syncthreads();
      commit();
      __syncthreads();
    }
  }
  for (; c0 < p.n; c0 += TS_KS) {
    const bool more = c0 + TS_KS < p.n;
    if (more) fetch(c0 + TS_KS);
#pragma unroll
    for (int k4 = 0; k4 < TS_KS / 4; ++k4) {
      double bq[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bq[j] = fb[4 * k4 * KP + 16 * j];
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        const double a = fa[64 * b * TS_SA + 4 * k4];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[b][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[j], acc[b][j], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) commit();
    __syncthreads();
  }
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long row = row0 + 64 * b + 16 * wid + l4 + 4 * c;
        const int col = 16 * j + l15;
        if (row < p.N && col < p.k) {
          if (p.nt_y) __builtin_nontemporal_store(acc[b][j][c], &p.Y[row * p.ldy + col]);
          else p.Y[row * p.ldy + col] = acc[b][j][c];
        }
      }
}

// Round 3, an alternative kept behind ROMTIME_TS_FLAGS & 4 - it measured SLOWER than the staged kernel (1.095 vs 1.04 ms at
// 1e6 x 512 -> 40, tools/probes/ts_ab.py, profiles/r03_tallskinny_ab.txt; non-temporal loads of X / stores of Y change
// nothing either: 1.03-1.04 ms): X never touches LDS.  The contraction index of an MFMA is a dummy: lane (l15, l4) of a 16x16x4 step may
// supply ANY k as long as the B operand of the same lane supplies the same one.  So a lane loads four CONSECUTIVE
// doubles of "its" row (one 32-byte load; the four l4 groups of a row together read 128 contiguous bytes) and feeds them
// to four successive MFMA steps, while the B operand is read from the T stage in LDS at row 16 jb + 4 l4 + step instead
// of 4 step + l4.  Against the staged kernel a stage of 32 columns loses 8 ds_write_b128 and 16 ds_read_b64 per thread
// (X never goes through LDS), their address arithmetic - all of it VALU work that an FP64 MFMA cannot overlap with - and
// one of the two barriers (the T stages are double buffered; T is the only thing the waves of a workgroup share).
template <int NT, int RB>
__global__ __launch_bounds__(TS_THREADS, RB == 2 ? (NT <= 4 ? 3 : 1) : (NT <= 4 ? 4 : 2)) void tallskinny_direct_kernel(const TsParams p) {
  constexpr int TS_BM = 64 * RB;
  constexpr int KP = 16 * NT;                                 // padded output width
  constexpr int TL = (TS_KS * KP / 2 + TS_THREADS - 1) / TS_THREADS;  // d2 loads of T per thread and stage
  constexpr int JB = TS_KS / 16;                              // 16-column blocks of a stage: one d4 load each
  __shared__ __attribute__((aligned(16))) double sT[2][TS_KS * KP];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long row0 = (long)blockIdx.x * TS_BM;

  d4 acc[RB][NT];
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[b][j] = d4{0.0, 0.0, 0.0, 0.0};

  // per-thread byte offsets, computed once; stage bases are wave-uniform and advance by scalar adds
  unsigned xoff[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) xoff[b] = (unsigned)(((long)(64 * b + 16 * wid + l15) * p.ldx + 4 * l4) * 8);
  unsigned toff[TL];
  bool tuse[TL];
  d2 tr[TL];
#pragma unroll
  for (int i = 0; i < TL; ++i) {
    const int q = tid + TS_THREADS * i, kk = q / (KP / 2), j = 2 * (q % (KP / 2));
    toff[i] = (unsigned)(((long)kk * p.ldt + j) * 8);
    tuse[i] = (kk < TS_KS) && (j + 1 < p.k);                                        // padded columns stay zero
    tr[i] = d2{0.0, 0.0};
  }
  const char* gx = reinterpret_cast<const char*>(p.X + row0 * p.ldx);
  const char* gt = reinterpret_cast<const char*>(p.T);
  const long tstage = (long)TS_KS * p.ldt * 8;
  auto commit = [&](double* dst) {
#pragma unroll
    for (int i = 0; i < TL; ++i) {
      const int q = tid + TS_THREADS * i, kk = q / (KP / 2), j = 2 * (q % (KP / 2));
      if (kk < TS_KS) *reinterpret_cast<d2*>(&dst[kk * KP + j]) = tr[i];
    }
  };
  d4 xa[RB][JB], xn[RB][JB];
  // stage 0
#pragma unroll
  for (int i = 0; i < TL; ++i)
    if (tuse[i]) tr[i] = *reinterpret_cast<const d2*>(gt + toff[i]);
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) xa[b][jb] = *reinterpret_cast<const d4*>(gx + xoff[b] + jb * 128);
  commit(sT[0]);
  __syncthreads();
  const int nstage = p.n / TS_KS;
  const double* fb0 = sT[0] + 4 * l4 * KP + l15;   // B operand of step st of block jb: row 16 jb + 4 l4 + st, column 16 j + l15
  for (int s = 0; s < nstage; ++s) {
    const bool more = s + 1 < nstage;
    if (more) {
      gx += TS_KS * 8;
      gt += tstage;
#pragma unroll
      for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int jb = 0; jb < JB; ++jb) xn[b][jb] = *reinterpret_cast<const d4*>(gx + xoff[b] + jb * 128);
#pragma unroll
      for (int i = 0; i < TL; ++i)
        if (tuse[i]) tr[i] = *reinterpret_cast<const d2*>(gt + toff[i]);
    }
    const double* fb = fb0 + (s & 1) * (TS_KS * KP);
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        double bq[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bq[j] = fb[(16 * jb + st) * KP + 16 * j];
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[b][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[b][jb][st], bq[j], acc[b][j], 0, 0, 0);
      }
    if (more) {
      commit(sT[(s + 1) & 1]);   // the other buffer: everybody left it at the barrier that ended stage s - 1
#pragma unroll
      for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int jb = 0; jb < JB; ++jb) xa[b][jb] = xn[b][jb];
    }
    __syncthreads();
  }
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long row = row0 + 64 * b + 16 * wid + l4 + 4 * c;
        const int col = 16 * j + l15;
        if (col < p.k) p.Y[row * p.ldy + col] = acc[b][j][c];
      }
}

}  // namespace

// RT_ERR_UNSUPPORTED: shape outside this kernel's range (the caller uses the generic GEMM).
int rt_tallskinny(rt_ctx* ctx, const double* X, int64_t ldx, const double* T, int64_t ldt, int64_t N, int64_t n,
                  int64_t k, double* Y, int64_t ldy) {
  if (k > 128 || n < 2 * TS_KS || N < 64L * ctx->num_cus) return RT_ERR_UNSUPPORTED;
  static const int ts_flags = [] { const char* e = getenv("ROMTIME_TS_FLAGS"); return e ? atoi(e) : 0; }();
  TsParams p{X, T, Y, (long)N, (long)ldx, (long)ldt, (long)ldy, (int)n, (int)k, ts_flags & 1, (ts_flags >> 3) & 1, (ts_flags >> 4) & 1};
  const int nt = (int)((k + 15) / 16);
  // two 16-row blocks per wave when the 128-row workgroups still fill the chip a few times over and the accumulators fit
  const int rb = (nt <= 4 && !(ts_flags & 2) && N >= 128L * 4 * ctx->num_cus) ? 2 : 1;
  const int bm = 64 * rb;
  const unsigned grid = (unsigned)((N + bm - 1) / bm);
  if (ctx->profile) {
    if (!ctx->ev0) {
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev0));
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev1));
    }
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  }
  // the direct form needs whole workgroups, whole stages and 32-byte aligned row segments; one partial workgroup at the
  // end of X goes through the staged kernel in a launch of its own
  const bool direct = (ts_flags & 4) && rb == 2 && (n % TS_KS == 0) && ((k & 1) == 0) && (ldx % 4 == 0) && (ldt % 2 == 0) &&
                      ((reinterpret_cast<size_t>(X) & 31) == 0) && ((reinterpret_cast<size_t>(T) & 15) == 0) &&
                      ((long)bm * ldx * 8 < (1L << 31)) && ((long)TS_KS * ldt * 8 < (1L << 31));
  const unsigned grid_direct = direct ? (unsigned)(N / bm) : 0u;
  TsParams ptail = p;
  if (direct) {
    const long done = (long)grid_direct * bm;
    ptail.X = X + done * ldx; ptail.Y = Y + done * ldy; ptail.N = N - done;
  }
#define TS_LAUNCH(NT_)                                                                                                  \
  if (direct) {                                                                                                         \
    hipLaunchKernelGGL((tallskinny_direct_kernel<(NT_ <= 4 ? NT_ : 4), 2>), dim3(grid_direct), dim3(TS_THREADS), 0, ctx->stream, p); \
    if (ptail.N > 0) hipLaunchKernelGGL((tallskinny_kernel<(NT_ <= 4 ? NT_ : 4), 2>), dim3(1), dim3(TS_THREADS), 0, ctx->stream, ptail); \
  } else if (rb == 2) hipLaunchKernelGGL((tallskinny_kernel<(NT_ <= 4 ? NT_ : 4), 2>), dim3(grid), dim3(TS_THREADS), 0, ctx->stream, p); \
  else hipLaunchKernelGGL((tallskinny_kernel<NT_, 1>), dim3(grid), dim3(TS_THREADS), 0, ctx->stream, p)
  switch (nt) {
    case 1: TS_LAUNCH(1); break;
    case 2: TS_LAUNCH(2); break;
    case 3: TS_LAUNCH(3); break;
    case 4: TS_LAUNCH(4); break;
    case 5: TS_LAUNCH(5); break;
    case 6: TS_LAUNCH(6); break;
    case 7: TS_LAUNCH(7); break;
    default: TS_LAUNCH(8); break;
  }
#undef TS_LAUNCH
  RT_HIP_CHECK(ctx, hipGetLastError());
  if (ctx->profile) {
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
  }
  ctx->last_grid = grid; ctx->last_splits = 1; ctx->last_tile = bm * 1000 + 16 * nt;
  return RT_OK;
}
