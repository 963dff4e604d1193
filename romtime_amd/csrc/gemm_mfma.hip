// FP64 matrix-core GEMM with the contraction on the long (DoF) axis.
//
//   C(i,j) = sum_k A(k,i) * B(k,j)
//
// One kernel serves the snapshot Gram matrix G = X^T X (pod.py:38, the O(N n^2) part of
// dgesvd), the reduced projections V^T (A V) / V^T Phi / V^T f (utils.py:112, deim.py:509,
// rom.py:156) and -- with the roles of the axes swapped by the strides -- the tall-skinny
// products X T (POD back-projection) and V u_N (rom.py:111-112).
//
// CDNA4 mapping: 256-thread workgroups = 4 waves in a 2x2 arrangement; each wave owns an
// (16 MT) x (16 NT) block of v_mfma_f64_16x16x4_f64 accumulators; operand panels are staged
// HBM -> registers -> LDS (double buffered, one barrier per 16-deep K stage) in whichever of
// two bank-conflict-free images matches the operand's contiguous axis, so neither layout
// needs a transpose.  The contraction is split over workgroups; partial tiles go to a slab
// and a second kernel sums them in a fixed order (bitwise reproducible, no atomics).
// blockIdx -> (split, tile) is XCD-aware: the tiles of one K-slice are dealt to one XCD so
// the slice of X they share is fetched from HBM once and re-read from that XCD's L2.
#include "common.h"

#include "gemm_panel.h"

using namespace rtk;

namespace {

// SKINNY: the 4 waves are stacked along M (each 16 MT rows x 16 NT columns) instead of 2 x 2, so an
// output of 40 columns costs 3 MFMA tiles per row block, not the 4 of a 64-wide 2 x 2 tile; the B panel
// is still loaded 64 wide (columns past Nn are zero-filled without touching memory).
template <int MT, int NT, bool KCA, bool KCB, bool SKINNY = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_f64_mfma_kernel(const GemmParams p) {
  constexpr int BM = SKINNY ? 64 * MT : 32 * MT;
  constexpr int BN = SKINNY ? 16 * NT : 32 * NT;       // output columns covered by a tile
  constexpr int BNP = SKINNY ? 64 : 32 * NT;            // width of the staged B panel
  using PA = Panel<BM, KCA>;
  using PB = Panel<BNP, KCB>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sA0 = smem;
  double* sA1 = smem + PA::LDS;
  double* sB0 = smem + 2 * PA::LDS;
  double* sB1 = smem + 2 * PA::LDS + PB::LDS;

  const int tid = threadIdx.x;
  int s, t;
  if (p.splits == 1) {
    s = 0;
    t = blockIdx.x;
  } else {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    s = (j / p.ntiles) * 8 + x;
    t = j % p.ntiles;
    if (s >= p.splits) return;
  }
  int tm, tn;
  if (p.symmetric) {
    tm = 0;
    int rem = t, rowlen = p.tiles_n;
    while (rem >= rowlen) {
      rem -= rowlen;
      ++tm;
      --rowlen;
    }
    tn = tm + rem;
  } else {
    tm = t / p.tiles_n;
    tn = t % p.tiles_n;
  }
  const bool diag = p.symmetric && (tm == tn);
  const long m0 = (long)tm * BM, n0 = (long)tn * BN;
  const long kbeg = (long)s * p.k_per_split;
  const long kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const int nstages = (kend > kbeg) ? (int)((kend - kbeg + KB - 1) / KB) : 0;

  const int lane = tid & 63, wid = tid >> 6;
  const int wm = SKINNY ? wid : (wid >> 1), wn = SKINNY ? 0 : (wid & 1);
  const int l15 = lane & 15, l4 = lane >> 4;

  d4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  d2 ra[PA::NL], rb[PB::NL];
  if (nstages > 0) {
    PA::load(ra, p.A, p.a_ks, p.a_ms, kbeg, kend, m0, p.M, p.vecA, tid);
    if (!diag) PB::load(rb, p.B, p.b_ks, p.b_ns, kbeg, kend, n0, p.Nn, p.vecB, tid);
    PA::store(ra, sA0, tid);
    if (!diag) PB::store(rb, sB0, tid);
  }
  __syncthreads();

  for (int st = 0; st < nstages; ++st) {
    const double* cA = (st & 1) ? sA1 : sA0;
    const double* cB = diag ? cA : ((st & 1) ? sB1 : sB0);
    const bool more = (st + 1 < nstages);
    if (more) {
      const long k0 = kbeg + (long)(st + 1) * KB;
      PA::load(ra, p.A, p.a_ks, p.a_ms, k0, kend, m0, p.M, p.vecA, tid);
      if (!diag) PB::load(rb, p.B, p.b_ks, p.b_ns, k0, kend, n0, p.Nn, p.vecB, tid);
    }
#pragma unroll
    for (int k4 = 0; k4 < KB / 4; ++k4) {
      double a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = PA::frag(cA, wm * 16 * MT + i * 16, k4, l15, l4);
      if (diag) {
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = PA::frag(cB, wn * 16 * NT + j * 16, k4, l15, l4);
      } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = PB::frag(cB, wn * 16 * NT + j * 16, k4, l15, l4);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      PA::store(ra, (st & 1) ? sA0 : sA1, tid);
      if (!diag) PB::store(rb, (st & 1) ? sB0 : sB1, tid);
    }
    __syncthreads();
  }

  // C/D map of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
  double* Cs = p.C + (long)s * p.c_split_stride;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * 16 * MT + i * 16 + l4 + 4 * r;
        const long col = n0 + wn * 16 * NT + j * 16 + l15;
        if (row < p.M && col < p.Nn) {
          double* dst = Cs + row * p.c_rs + col * p.c_cs;
          const double v = p.c_alpha * acc[i][j][r];  // alpha == 1 for slabs and plain products: exact
          *dst = (p.c_beta != 0.0) ? fma(p.c_beta, *dst, v) : v;
        }
      }
}

// Fixed-order sum of the split slabs (row-major M x Nn, ld = Nn).  symmetric: element (i,j)
// with i > j is read from (j,i), which also makes the result exactly symmetric.
__global__ void gemm_reduce_kernel(const double* __restrict__ slab, long split_stride, int splits, double* C,
                                   long c_rs, long c_cs, long M, long Nn, int symmetric, double alpha, double beta) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * Nn) return;
  const long i = idx / Nn, j = idx % Nn;
  long si = i, sj = j;
  if (symmetric && i > j) {
    si = j;
    sj = i;
  }
  const double* src = slab + si * Nn + sj;
  double sum = 0.0;
  for (int s = 0; s < splits; ++s) sum += src[(long)s * split_stride];
  double* dst = C + i * c_rs + j * c_cs;
  const double v = alpha * sum;
  *dst = (beta != 0.0) ? fma(beta, *dst, v) : v;
}

template <int MT, int NT, bool KCA, bool KCB, bool SKINNY = false>
int launch(rt_ctx* ctx, const GemmParams& p, int grid) {
  constexpr int BM = SKINNY ? 64 * MT : 32 * MT, BNP = SKINNY ? 64 : 32 * NT;
  constexpr size_t lds = sizeof(double) * 2 * (Panel<BM, KCA>::LDS + Panel<BNP, KCB>::LDS);
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&gemm_f64_mfma_kernel<MT, NT, KCA, KCB, SKINNY>), (int)lds));
  hipLaunchKernelGGL((gemm_f64_mfma_kernel<MT, NT, KCA, KCB, SKINNY>), dim3(grid), dim3(NTHREADS), lds, ctx->stream, p);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

template <int MT, int NT>
int launch_layout(rt_ctx* ctx, const GemmParams& p, int grid, bool kca, bool kcb) {
  if (kca && kcb) return launch<MT, NT, true, true>(ctx, p, grid);
  if (kca && !kcb) return launch<MT, NT, true, false>(ctx, p, grid);
  if (!kca && kcb) return launch<MT, NT, false, true>(ctx, p, grid);
  return launch<MT, NT, false, false>(ctx, p, grid);
}

template <int MT>
int launch_nt(rt_ctx* ctx, const GemmParams& p, int grid, int nt, bool kca, bool kcb) {
  switch (nt) {
    case 1: return launch_layout<MT, 1>(ctx, p, grid, kca, kcb);
    case 2: return launch_layout<MT, 2>(ctx, p, grid, kca, kcb);
    case 3: return launch_layout<MT, 3>(ctx, p, grid, kca, kcb);
    default: return launch_layout<MT, 4>(ctx, p, grid, kca, kcb);
  }
}

template <int NT>
int launch_skinny(rt_ctx* ctx, const GemmParams& p, int grid, bool kca, bool kcb) {
  if (kca && kcb) return launch<2, NT, true, true, true>(ctx, p, grid);
  if (kca && !kcb) return launch<2, NT, true, false, true>(ctx, p, grid);
  if (!kca && kcb) return launch<2, NT, false, true, true>(ctx, p, grid);
  return launch<2, NT, false, false, true>(ctx, p, grid);
}

int tile_units(long extent) {  // tile extent in units of 32 (1..4)
  if (extent >= 128) return 4;
  return (int)((extent + 31) / 32);
}

}  // namespace

int rt_gemm_strided(rt_ctx* ctx, const double* A, int64_t a_ks, int64_t a_ms, const double* B, int64_t b_ks,
                    int64_t b_ns, int64_t K, int64_t M, int64_t Nn, double* C, int64_t c_rs, int64_t c_cs,
                    bool symmetric, bool allow_split, double alpha, double beta) {
  RT_ARG_CHECK(ctx, A && B && C);
  RT_ARG_CHECK(ctx, K >= 0 && M >= 1 && Nn >= 1);
  RT_ARG_CHECK(ctx, a_ks == 1 || a_ms == 1 || M == 1 || K == 1);
  RT_ARG_CHECK(ctx, b_ks == 1 || b_ns == 1 || Nn == 1 || K == 1);
  if (symmetric) RT_ARG_CHECK(ctx, A == B && M == Nn && a_ks == b_ks && a_ms == b_ns);

  const bool kca = (a_ks == 1) && !(a_ms == 1 && M > 1);
  const bool kcb = (b_ks == 1) && !(b_ns == 1 && Nn > 1);
  int mt = tile_units(M), nt = tile_units(Nn);
  if (symmetric) nt = mt;
  // A product too small to give every CU a tile is latency-bound: narrower tiles, more workgroups (the hyper-reduced
  // sweep's 64 x 6400 x 280 expansion: 50 tiles of 64 x 128 took 33 us, 200 of 64 x 32 take a third of that)
  if (!symmetric)
    while (nt > 1 && ((M + 32 * mt - 1) / (32 * mt)) * ((Nn + 32 * nt - 1) / (32 * nt)) < ctx->num_cus) --nt;
  // tall output with <= 64 columns (POD back-projection, lift of a batch of reduced vectors)
  const bool skinny = !symmetric && Nn <= 64 && M >= 256;
  const int snt = (int)((Nn + 15) / 16);
  const int BM = skinny ? 128 : 32 * mt, BN = skinny ? 16 * snt : 32 * nt;

  GemmParams p;
  p.A = A; p.a_ks = a_ks; p.a_ms = a_ms;
  p.B = B; p.b_ks = b_ks; p.b_ns = b_ns;
  p.K = K; p.M = M; p.Nn = Nn;
  p.tiles_m = (int)((M + BM - 1) / BM);
  p.tiles_n = (int)((Nn + BN - 1) / BN);
  p.symmetric = symmetric ? 1 : 0;
  p.ntiles = symmetric ? p.tiles_m * (p.tiles_m + 1) / 2 : p.tiles_m * p.tiles_n;
  // 16-byte vector loads need the pair of elements contiguous and 16-B aligned for every (k, m)
  auto vec_ok = [](const double* P, long ks, long ms, bool kc) {
    const long other = kc ? ms : ks;
    return (((uintptr_t)P & 15) == 0) && (other % 2 == 0) && ((kc ? ks : ms) == 1);
  };
  p.vecA = vec_ok(A, a_ks, a_ms, kca) ? 1 : 0;
  p.vecB = vec_ok(B, b_ks, b_ns, kcb) ? 1 : 0;

  const int slots = 2 * ctx->num_cus;
  int splits = 1;
  if (allow_split && p.ntiles < slots) {
    splits = slots / p.ntiles;
    if (splits >= 8) splits &= ~7;
    const long min_k = 8 * KB;  // do not split below 128 rows per workgroup
    long kps = (K + splits - 1) / splits;
    if (kps < min_k) kps = min_k;
    kps = (kps + KB - 1) / KB * KB;
    splits = (int)((K + kps - 1) / kps);
    if (splits < 1) splits = 1;
    p.k_per_split = kps;
  } else {
    p.k_per_split = (K + KB - 1) / KB * KB;
    if (p.k_per_split == 0) p.k_per_split = KB;
  }
  p.splits = splits;

  const bool use_slab = (splits > 1) || symmetric;
  if (use_slab) {
    void* slab = nullptr;
    const size_t bytes = sizeof(double) * (size_t)splits * (size_t)M * (size_t)Nn;
    int rc = rt_scratch(ctx, bytes, &slab);
    if (rc != RT_OK) return rc;
    p.C = static_cast<double*>(slab);
    p.c_rs = Nn; p.c_cs = 1; p.c_split_stride = M * Nn;
    p.c_alpha = 1.0; p.c_beta = 0.0;  // slabs hold plain partial products; alpha / beta are applied by the reduction
  } else {
    p.C = C; p.c_rs = c_rs; p.c_cs = c_cs; p.c_split_stride = 0;
    p.c_alpha = alpha; p.c_beta = beta;
  }
  const int grid = (splits == 1) ? p.ntiles : 8 * ((splits + 7) / 8) * p.ntiles;
  ctx->last_grid = grid; ctx->last_splits = splits; ctx->last_tile = BM * 1000 + BN;

  int rc = RT_OK;
  if (ctx->profile) {
    if (!ctx->ev0) {
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev0));
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev1));
    }
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  }
  if (skinny) {
    switch (snt) {
      case 1: rc = launch_skinny<1>(ctx, p, grid, kca, kcb); break;
      case 2: rc = launch_skinny<2>(ctx, p, grid, kca, kcb); break;
      case 3: rc = launch_skinny<3>(ctx, p, grid, kca, kcb); break;
      default: rc = launch_skinny<4>(ctx, p, grid, kca, kcb); break;
    }
  } else
  switch (mt) {
    case 1: rc = launch_nt<1>(ctx, p, grid, nt, kca, kcb); break;
    case 2: rc = launch_nt<2>(ctx, p, grid, nt, kca, kcb); break;
    case 3: rc = launch_nt<3>(ctx, p, grid, nt, kca, kcb); break;
    default: rc = launch_nt<4>(ctx, p, grid, nt, kca, kcb); break;
  }
  if (rc != RT_OK) return rc;
  if (ctx->profile) {
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
  }
  if (use_slab) {
    const long total = M * Nn;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       p.C, p.c_split_stride, splits, C, (long)c_rs, (long)c_cs, (long)M, (long)Nn,
                       symmetric ? 1 : 0, alpha, beta);
    RT_HIP_CHECK(ctx, hipGetLastError());
  }
  return RT_OK;
}
