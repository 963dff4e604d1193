// Fused reduced projection  A_N[b] = V^T (A_b V)  for B value-vectors on one CSR pattern
// (project_csr, utils.py:96-113; MDEIM.project_basis, mdeim.py:153-192; the online direct path,
// rom.py:149-153) without materialising A_b V in HBM.
//
// One workgroup owns one (value-vector b, DoF-range) pair.  Per 16-row stage it
//   * finds the column window of the stage's rows and, when it spans <= 64 DoFs (banded FE operators),
//     copies those V rows into LDS once; the stage's own 16 rows inside it are the A operand,
//   * builds the 16 rows of (A_b V) on the fly: row k = sum_e a_b[e] V[col_e][:], read from the LDS
//     window (or gathered through L1/L2 when the window is too wide), written into the LDS image of
//     the B operand,
//   * accumulates the r x r product on the FP64 matrix cores: the ceil(r/16)^2 MFMA tiles are dealt
//     round-robin to the 8 waves, operands read from LDS at run-time offsets, so r = 80 costs 25
//     tiles (not the 36 of a padded 96 x 96 tile).
// Partial r x r blocks of the DoF-ranges go to slabs and are summed in a fixed order.
// HBM traffic: the value vectors once (8 nnz B bytes) + V once; the unfused path moved 16 N r B more.
#include "common.h"

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PK = 32;        // rows per stage
constexpr int PT = 512;       // threads (8 waves)
constexpr int RMAX = 128;

struct ProjParams {
  const long* indptr;
  const long* indices;
  const double* data;
  long d_es, d_bs;
  const double* V;
  long ldv;
  double* slab;        // [B][S][r*r]
  long N, k_per_split;
  int r, B, S, tr, stride;
  const int* win;      // [2 * stages]: column window (lo, hi) of every 16-row stage, hi = -1: not windowed
};

constexpr int WROWS = 48;     // V rows kept in LDS per stage (banded FE operators touch ~PK + 2*bandwidth)
constexpr int EMAX = PT;      // entries of one stage staged through LDS (one per thread)
constexpr int WREG = WROWS / 4;  // window doubles per thread in flight: column tid % 128, rows tid / 128 + 4 i

// Column window [lo, hi] of every 16-row stage; hi = -1 marks "do not window" (too wide / too many entries).
__global__ void project_windows_kernel(const long* __restrict__ indptr, const long* __restrict__ indices, long N,
                                       int* __restrict__ win) {
  const long st = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long k0 = st * PK;
  if (k0 >= N) return;
  const long k1 = (k0 + PK < N) ? k0 + PK : N;
  int lo = (int)k0, hi = (int)(k1 - 1);
  const long e0 = indptr[k0], e1 = indptr[k1];
  for (long e = e0; e < e1; ++e) {
    const int c = (int)indices[e];
    lo = min(lo, c);
    hi = max(hi, c);
  }
  const bool ok = (hi - lo + 1 <= WROWS) && (e1 - e0 <= EMAX);
  win[2 * st] = ok ? lo : (int)k0;
  win[2 * st + 1] = ok ? hi : -1;
}

// Software-pipelined: while stage s is computed out of LDS, the entries (column, value) and the V-row
// window of stage s+1 are in flight into registers; they are written to LDS after the MFMAs.
template <int PACC>  // MFMA tiles per wave: ceil(ceil(r/16)^2 / 8)
__global__ __launch_bounds__(PT, PACC <= 4 ? 4 : 2) void project_fused_kernel(const ProjParams p) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* sW = sm;                             // [WROWS][stride]  window of V rows
  double* sB = sm + (size_t)WROWS * p.stride;  // [PK][stride]
  double* sVal = sB + (size_t)PK * p.stride;   // [EMAX] values of the stage's entries
  int* sCol = reinterpret_cast<int*>(sVal + EMAX);  // [EMAX] window-relative columns
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int b = blockIdx.x % p.B, s = blockIdx.x / p.B;
  const int r = p.r, tr = p.tr, ntile = tr * tr, stride = p.stride;
  const long kbeg = (long)s * p.k_per_split;
  const long kend = (kbeg + p.k_per_split < p.N) ? kbeg + p.k_per_split : p.N;
  const double* dat = p.data + (long)b * p.d_bs;

  d4 acc[PACC];
  int ti[PACC], tj[PACC];
#pragma unroll
  for (int q = 0; q < PACC; ++q) {
    acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    const int t = q * 8 + wid;
    ti[q] = (t < ntile) ? 16 * (t / tr) : -1;
    tj[q] = (t < ntile) ? 16 * (t % tr) : 0;
  }
  const int rp = tr * 16;                  // padded width
  constexpr int CG = PT / PK;              // column groups of the B-operand mapping (PK rows x CG groups)
  constexpr int NCOL = RMAX / CG;          // columns per thread: jg, jg + CG, ...
  const int kk = tid / CG, jg = tid % CG;

  // registers of the stage in flight
  const int wrow = tid >> 7, wcol = tid & 127;
  double wreg[WREG];
  double vreg = 0.0;
  int creg = 0;
  int n_lo = 0, n_hi = -1;

  auto fetch = [&](long k0) {  // issue the loads of stage k0 (no LDS access)
    const long st = k0 / PK;
    n_lo = p.win[2 * st];
    n_hi = p.win[2 * st + 1];
    const long k1 = (k0 + PK < kend) ? k0 + PK : kend;
    const int nrow = (n_hi >= 0) ? (n_hi - n_lo + 1) : (int)(k1 - k0);
    const double* vsrc = p.V + (long)(n_lo + wrow) * p.ldv + wcol;
#pragma unroll
    for (int i = 0; i < WREG; ++i)  // thread -> column wcol, rows wrow + 4 i
      wreg[i] = (wrow + 4 * i < nrow && wcol < r) ? vsrc[(long)(4 * i) * p.ldv] : 0.0;
    if (n_hi >= 0) {
      const long e = p.indptr[k0] + tid;
      const bool v = e < p.indptr[k1];
      vreg = v ? dat[e * p.d_es] : 0.0;
      creg = v ? (int)p.indices[e] - n_lo : 0;
    }
  };
  auto commit = [&]() {  // registers -> LDS
    if (wcol < rp) {
#pragma unroll
      for (int i = 0; i < WREG; ++i) sW[(wrow + 4 * i) * stride + wcol] = wreg[i];
    }
    sVal[tid] = vreg;
    sCol[tid] = creg;
  };

  if (kbeg < kend) {
    fetch(kbeg);
    commit();
  }
  int c_lo = n_lo, c_hi = n_hi;
  __syncthreads();

  for (long k0 = kbeg; k0 < kend; k0 += PK) {
    const bool more = (k0 + PK < kend);
    if (more) fetch(k0 + PK);
    const bool windowed = (c_hi >= 0);
    const int abase = (int)(k0 - c_lo);  // LDS row of DoF k0 (0 when not windowed)
    // B operand: rows of A_b V
    {
      const long k = k0 + kk;
      double o[NCOL];
#pragma unroll
      for (int c = 0; c < NCOL; ++c) o[c] = 0.0;
      if (k < kend) {
        const long e0 = p.indptr[k], e1 = p.indptr[k + 1];
        if (windowed) {
          const int base = (int)(e0 - p.indptr[k0]);
          const int cnt = (int)(e1 - e0);
          for (int q = 0; q < cnt; ++q) {
            const double a = sVal[base + q];
            const double* vr = sW + sCol[base + q] * stride;
#pragma unroll
            for (int c = 0; c < NCOL; ++c)
              if (jg + CG * c < rp) o[c] = fma(a, vr[jg + CG * c], o[c]);
          }
        } else {
          for (long e = e0; e < e1; ++e) {
            const double a = dat[e * p.d_es];
            const double* vr = p.V + p.indices[e] * p.ldv;
#pragma unroll
            for (int c = 0; c < NCOL; ++c)
              if (jg + CG * c < r) o[c] = fma(a, vr[jg + CG * c], o[c]);
          }
        }
      }
      double* row = sB + kk * stride;
#pragma unroll
      for (int c = 0; c < NCOL; ++c)
        if (jg + CG * c < rp) row[jg + CG * c] = o[c];
    }
    __syncthreads();
    const double* cA = sW + abase * stride;
#pragma unroll 1  // keeps the LDS operand loads of one k-step (not four) in flight: VGPRs <= 128, 2 workgroups per CU
    for (int k4 = 0; k4 < PK / 4; ++k4) {
      const int rowoff = (k4 * 4 + l4) * stride + l15;
#pragma unroll
      for (int q = 0; q < PACC; ++q) {
        if (ti[q] >= 0) {  // wave-uniform
          const double a = cA[rowoff + ti[q]];
          const double bb = sB[rowoff + tj[q]];
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[q], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) {
      commit();
      c_lo = n_lo;
      c_hi = n_hi;
    }
    __syncthreads();
  }
  double* out = p.slab + ((long)b * p.S + s) * ((long)r * r);
#pragma unroll
  for (int q = 0; q < PACC; ++q) {
    if (ti[q] < 0) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = ti[q] + l4 + 4 * c, col = tj[q] + l15;
      if (row < r && col < r) out[(long)row * r + col] = acc[q][c];
    }
  }
}

__global__ void project_reduce_kernel(const double* __restrict__ slab, int S, long rr, long total,
                                      double* __restrict__ AN) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long b = idx / rr, e = idx % rr;
  const double* src = slab + (b * S) * rr + e;
  double sum = 0.0;
  for (int s = 0; s < S; ++s) sum += src[(long)s * rr];
  AN[idx] = sum;
}

}  // namespace

// Returns RT_ERR_UNSUPPORTED for r > 128 (the caller then uses the unfused path).
int rt_project_fused(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, const double* data, int64_t d_es,
                     int64_t d_bs, int64_t B, int64_t N, const double* V, int64_t ldv, int64_t r, double* AN) {
  if (r > RMAX) return RT_ERR_UNSUPPORTED;
  ProjParams p;
  p.indptr = reinterpret_cast<const long*>(indptr);
  p.indices = reinterpret_cast<const long*>(indices);
  p.data = data; p.d_es = d_es; p.d_bs = d_bs;
  p.V = V; p.ldv = ldv; p.N = N; p.r = (int)r; p.B = (int)B;
  p.tr = (int)((r + 15) / 16);
  const int rp = p.tr * 16;
  p.stride = ((rp + 31) / 32) * 32 + 16;  // == 16 (mod 32): conflict-free ds_read_b64 of the MFMA operands
  const long slots = 4L * ctx->num_cus;   // a few workgroups per CU so that gather and MFMA phases overlap
  long S = (slots + B - 1) / B;
  const long stages = (N + PK - 1) / PK;
  if (S > stages / 8) S = stages / 8;     // at least 8 stages per workgroup
  if (S < 1) S = 1;
  p.k_per_split = ((N + S - 1) / S + PK - 1) / PK * PK;
  S = (N + p.k_per_split - 1) / p.k_per_split;
  p.S = (int)S;
  void* slab = nullptr;
  const size_t slab_bytes = (sizeof(double) * (size_t)B * S * r * r + 255) / 256 * 256;
  int rc = rt_scratch(ctx, slab_bytes + sizeof(int) * 2 * (size_t)stages, &slab);
  if (rc != RT_OK) return rc;
  p.slab = static_cast<double*>(slab);
  int* win = reinterpret_cast<int*>(static_cast<char*>(slab) + slab_bytes);
  p.win = win;
  hipLaunchKernelGGL(project_windows_kernel, dim3((unsigned)((stages + 255) / 256)), dim3(256), 0, ctx->stream,
                     p.indptr, p.indices, (long)N, win);
  RT_HIP_CHECK(ctx, hipGetLastError());
  const size_t lds = sizeof(double) * ((size_t)(WROWS + PK) * p.stride + EMAX) + sizeof(int) * EMAX;
  static bool attr_set = false;
  if (!attr_set) {
    RT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&project_fused_kernel<2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    RT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&project_fused_kernel<4>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    RT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&project_fused_kernel<8>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    attr_set = true;
  }
  if (ctx->profile) {
    if (!ctx->ev0) {
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev0));
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev1));
    }
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  }
  const int tiles_per_wave = (p.tr * p.tr + 7) / 8;
  if (tiles_per_wave <= 2)
    hipLaunchKernelGGL(project_fused_kernel<2>, dim3((unsigned)(B * S)), dim3(PT), lds, ctx->stream, p);
  else if (tiles_per_wave <= 4)
    hipLaunchKernelGGL(project_fused_kernel<4>, dim3((unsigned)(B * S)), dim3(PT), lds, ctx->stream, p);
  else
    hipLaunchKernelGGL(project_fused_kernel<8>, dim3((unsigned)(B * S)), dim3(PT), lds, ctx->stream, p);
  RT_HIP_CHECK(ctx, hipGetLastError());
  if (ctx->profile) {
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
  }
  const long rr = r * r, total = B * rr;
  hipLaunchKernelGGL(project_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                     p.slab, (int)S, rr, total, AN);
  RT_HIP_CHECK(ctx, hipGetLastError());
  ctx->last_grid = B * S; ctx->last_splits = S; ctx->last_tile = rp * 1000 + rp;
  return RT_OK;
}
