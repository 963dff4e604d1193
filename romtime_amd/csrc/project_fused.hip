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

#include <utility>

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PK = 32;        // rows per stage
constexpr int PT = 512;       // threads (8 waves)
constexpr int RMAX = 128;
constexpr int WROWS = 48;     // V rows kept in LDS per stage (banded FE operators touch ~PK + 2*bandwidth)
constexpr int EMAX = PT;      // entries of one stage staged through LDS (one per thread)
constexpr int WREG = WROWS / 4;  // window doubles per thread in flight: column tid % 128, rows tid / 128 + 4 i
constexpr int MAXT = 8;       // MFMA tiles per wave at most (r = 128: 64 tiles on 8 waves)

// Everything a stage needs to start its loads, in one wave-uniform 32-byte record (one scalar load, issued a
// whole stage ahead): entry range [e0, e1) of the stage's rows and their column window [lo, hi];
// hi = -1 marks "do not window" (too wide / too many entries).
struct StageRec {
  long e0, e1;
  int lo, hi;
  int pad0, pad1;
};

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
  const StageRec* rec;  // [stages]
  // MFMA tiles of each wave, grouped in strips that share the A operand (same tile row): one byte (i << 4 | j)
  // per tile, 0xff = none
  unsigned long long tiles[8];
};

__global__ void project_stages_kernel(const long* __restrict__ indptr, const long* __restrict__ indices, long N,
                                      StageRec* __restrict__ rec) {
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
  StageRec o;
  o.e0 = e0;
  o.e1 = e1;
  o.lo = ok ? lo : (int)k0;
  o.hi = ok ? hi : -1;
  o.pad0 = o.pad1 = 0;
  rec[st] = o;
}

// Software-pipelined: while stage s is computed out of LDS, the entries (column, value), the per-row entry
// ranges and the V-row window of stage s+1 are in flight into registers (their addresses come from the stage
// record fetched during stage s-1, so no load waits on another load); they are written to LDS after the MFMAs.
template <int TR, int PACC>  // TR = ceil(r/16), PACC = most MFMA tiles any wave owns
__global__ __launch_bounds__(PT, TR <= 5 ? 4 : 2) void project_fused_kernel(const ProjParams p) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* sW = sm;                             // [WROWS][stride]  window of V rows
  double* sB = sm + (size_t)WROWS * p.stride;  // [PK][stride]
  double* sVal = sB + (size_t)PK * p.stride;   // [EMAX] values of the stage's entries
  int* sCol = reinterpret_cast<int*>(sVal + EMAX);  // [EMAX] window-relative columns
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int b = blockIdx.x % p.B, s = blockIdx.x / p.B;
  const int r = p.r, stride = p.stride;
  const long kbeg = (long)s * p.k_per_split;
  const long kend = (kbeg + p.k_per_split < p.N) ? kbeg + p.k_per_split : p.N;
  const double* dat = p.data + (long)b * p.d_bs;

  d4 acc[PACC];
  int ti[PACC], tj[PACC];
  const unsigned long long tl = p.tiles[wid];
  int ntl = 0;
#pragma unroll
  for (int q = 0; q < PACC; ++q) {
    acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    const unsigned t = (unsigned)(tl >> (8 * q)) & 0xffu;
    ti[q] = (int)(t >> 4) * 16;
    tj[q] = (int)(t & 15u) * 16;
    if (t != 0xffu) ntl = q + 1;
  }
  constexpr int rp = TR * 16;              // padded width
  constexpr int CG = PT / PK;              // column groups of the B-operand mapping (PK rows x CG groups)
  constexpr int NCOL = TR;                 // columns per thread: jg, jg + CG, ...
  static_assert(CG == 16, "one thread column per 16-wide tile column");
  const int kk = tid / CG, jg = tid % CG;

  // registers of the stage in flight
  const int wrow = tid >> 7, wcol = tid & 127;
  double wreg[WREG];
  double vreg = 0.0;
  int creg = 0, nb_lo = 0, nb_hi = 0;
  StageRec rn{};  // record of the stage in flight

  auto fetch = [&](long k0, const StageRec& rc) {  // issue the loads of stage k0 (no LDS access, no dependent load)
    rn = rc;
    const long k1 = (k0 + PK < kend) ? k0 + PK : kend;
    const int nrow = (rc.hi >= 0) ? (rc.hi - rc.lo + 1) : (int)(k1 - k0);
    const double* vsrc = p.V + (long)(rc.lo + wrow) * p.ldv + wcol;
#pragma unroll
    for (int i = 0; i < WREG; ++i)  // thread -> column wcol, rows wrow + 4 i
      wreg[i] = (wrow + 4 * i < nrow && wcol < r) ? vsrc[(long)(4 * i) * p.ldv] : 0.0;
    if (rc.hi >= 0) {
      const long e = rc.e0 + tid;
      const bool v = e < rc.e1;
      vreg = v ? dat[e * p.d_es] : 0.0;
      creg = v ? (int)p.indices[e] - rc.lo : 0;
      const long kr = k0 + kk;  // entry range of this thread's row, low words (the stage holds < 2^31 entries)
      const bool rv = kr < kend;
      nb_lo = rv ? (int)p.indptr[kr] : 0;
      nb_hi = rv ? (int)p.indptr[kr + 1] : 0;
    }
  };
  int c_lo = 0, c_hi = -1, row_base = 0, row_cnt = 0;
  auto commit = [&]() {  // registers -> LDS
    if (wcol < rp) {
#pragma unroll
      for (int i = 0; i < WREG; ++i) sW[(wrow + 4 * i) * stride + wcol] = wreg[i];
    }
    sVal[tid] = vreg;
    sCol[tid] = creg;
    c_lo = rn.lo;
    c_hi = rn.hi;
    row_base = nb_lo - (int)rn.e0;
    row_cnt = nb_hi - nb_lo;
  };

  const long st0 = kbeg / PK;
  StageRec r1{}, r2{};
  if (kbeg < kend) {
    r1 = p.rec[st0];
    if (kbeg + PK < kend) r2 = p.rec[st0 + 1];
    fetch(kbeg, r1);
    commit();
  }
  __syncthreads();

  long st = st0;
  for (long k0 = kbeg; k0 < kend; k0 += PK, ++st) {
    const bool more = (k0 + PK < kend);
    if (more) {
      r1 = r2;                                      // loaded one stage ago
      if (k0 + 2 * PK < kend) r2 = p.rec[st + 2];   // for the next iteration
      fetch(k0 + PK, r1);
    }
    const bool windowed = (c_hi >= 0);
    const int abase = (int)(k0 - c_lo);  // LDS row of DoF k0 (0 when not windowed)
    // B operand: rows of A_b V
    {
      const long k = k0 + kk;
      double o[NCOL];
#pragma unroll
      for (int c = 0; c < NCOL; ++c) o[c] = 0.0;
      if (windowed) {
        for (int q = 0; q < row_cnt; ++q) {
          const double a = sVal[row_base + q];
          const double* vr = sW + sCol[row_base + q] * stride + jg;
#pragma unroll
          for (int c = 0; c < NCOL; ++c) o[c] = fma(a, vr[CG * c], o[c]);
        }
      } else if (k < kend) {
        const long e0 = p.indptr[k], e1 = p.indptr[k + 1];
        for (long e = e0; e < e1; ++e) {
          const double a = dat[e * p.d_es];
          const double* vr = p.V + p.indices[e] * p.ldv;
#pragma unroll
          for (int c = 0; c < NCOL; ++c)
            if (jg + CG * c < r) o[c] = fma(a, vr[jg + CG * c], o[c]);
        }
      }
      double* row = sB + kk * stride + jg;
#pragma unroll
      for (int c = 0; c < NCOL; ++c) row[CG * c] = o[c];
    }
    __syncthreads();
    const double* cA = sW + abase * stride;
#pragma unroll 1  // keeps the LDS operand loads of one k-step (not four) in flight
    for (int k4 = 0; k4 < PK / 4; ++k4) {
      const int rowoff = (k4 * 4 + l4) * stride + l15;
      double a = 0.0;
#pragma unroll
      for (int q = 0; q < PACC; ++q) {
        if (q < ntl) {  // wave-uniform
          if (q == 0 || ti[q] != ti[q - 1]) a = cA[rowoff + ti[q]];  // strips share the A operand
          const double bb = sB[rowoff + tj[q]];
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[q], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) commit();
    __syncthreads();
  }
  double* out = p.slab + ((long)b * p.S + s) * ((long)r * r);
#pragma unroll
  for (int q = 0; q < PACC; ++q) {
    if (q >= ntl) continue;
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

// MFMA tiles -> waves.  Each tile row is cut into strips of <= 3 tiles that share the A operand (LDS reads per
// k-step: 1 + len instead of 2 len), strips are dealt longest-first to the least loaded wave, and the waves are
// ordered so that the pairs (w, w + 4), which share a SIMD, carry balanced MFMA work.
static void assign_tiles(ProjParams& p) {
  const int TR = p.tr;
  struct Strip { int i, j0, len; };
  Strip strips[8 * 3];
  int ns = 0;
  const int per_row = (TR + 2) / 3;
  for (int i = 0; i < TR; ++i) {
    int j0 = 0;
    for (int q = 0; q < per_row; ++q) {
      const int len = TR / per_row + (q < TR % per_row ? 1 : 0);
      strips[ns++] = Strip{i, j0, len};
      j0 += len;
    }
  }
  for (int a = 1; a < ns; ++a)  // stable insertion sort, longest first
    for (int c = a; c > 0 && strips[c].len > strips[c - 1].len; --c) std::swap(strips[c], strips[c - 1]);
  int load[8] = {0}, cnt[8] = {0};
  unsigned char li[8][MAXT], lj[8][MAXT];
  for (int a = 0; a < ns; ++a) {
    int w = 0;
    for (int c = 1; c < 8; ++c)
      if (load[c] < load[w]) w = c;
    for (int q = 0; q < strips[a].len; ++q) {
      li[w][cnt[w]] = (unsigned char)strips[a].i;
      lj[w][cnt[w]] = (unsigned char)(strips[a].j0 + q);
      ++cnt[w];
    }
    load[w] += strips[a].len;
  }
  int order[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  for (int a = 1; a < 8; ++a)
    for (int c = a; c > 0 && load[order[c]] > load[order[c - 1]]; --c) std::swap(order[c], order[c - 1]);
  for (int w = 0; w < 8; ++w) {
    const int src = (w < 4) ? order[w] : order[7 - (w - 4)];  // heaviest with lightest on one SIMD
    unsigned long long packed = 0;
    for (int q = 0; q < MAXT; ++q) {
      const unsigned long long t = q < cnt[src] ? (unsigned long long)((li[src][q] << 4) | lj[src][q]) : 0xffull;
      packed |= t << (8 * q);
    }
    p.tiles[w] = packed;
  }
}

size_t rt_project_stage_table_bytes(int64_t N) { return sizeof(StageRec) * (size_t)((N + PK - 1) / PK); }

int rt_project_stage_table(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, int64_t N, void* table) {
  const long stages = (N + PK - 1) / PK;
  hipLaunchKernelGGL(project_stages_kernel, dim3((unsigned)((stages + 255) / 256)), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const long*>(indptr), reinterpret_cast<const long*>(indices), (long)N,
                     static_cast<StageRec*>(table));
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

template <int TR, int PACC>
static int launch_fused(rt_ctx* ctx, const ProjParams& p, unsigned grid, size_t lds) {
  static bool attr_set = false;
  if (!attr_set) {
    RT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&project_fused_kernel<TR, PACC>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((project_fused_kernel<TR, PACC>), dim3(grid), dim3(PT), lds, ctx->stream, p);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

// Returns RT_ERR_UNSUPPORTED for r > 128 (the caller then uses the unfused path).  `stage_table` = the table of
// rt_project_stage_table for this pattern, or nullptr (built here, one small launch).
int rt_project_fused(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, const double* data, int64_t d_es,
                     int64_t d_bs, int64_t B, int64_t N, const double* V, int64_t ldv, int64_t r, double* AN,
                     const void* stage_table) {
  if (r > RMAX) return RT_ERR_UNSUPPORTED;
  ProjParams p;
  p.indptr = reinterpret_cast<const long*>(indptr);
  p.indices = reinterpret_cast<const long*>(indices);
  p.data = data; p.d_es = d_es; p.d_bs = d_bs;
  p.V = V; p.ldv = ldv; p.N = N; p.r = (int)r; p.B = (int)B;
  p.tr = (int)((r + 15) / 16);
  const int rp = p.tr * 16;
  p.stride = ((rp + 31) / 32) * 32 + 16;  // == 16 (mod 32): conflict-free ds_read_b64 of the MFMA operands
  assign_tiles(p);
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
  int rc = rt_scratch(ctx, slab_bytes + (stage_table ? 0 : rt_project_stage_table_bytes(N)), &slab);
  if (rc != RT_OK) return rc;
  p.slab = static_cast<double*>(slab);
  if (stage_table) {
    p.rec = static_cast<const StageRec*>(stage_table);
  } else {
    void* table = static_cast<char*>(slab) + slab_bytes;
    rc = rt_project_stage_table(ctx, indptr, indices, N, table);
    if (rc != RT_OK) return rc;
    p.rec = static_cast<const StageRec*>(table);
  }
  const size_t lds = sizeof(double) * ((size_t)(WROWS + PK) * p.stride + EMAX) + sizeof(int) * EMAX;
  if (ctx->profile) {
    if (!ctx->ev0) {
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev0));
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev1));
    }
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  }
  const unsigned grid = (unsigned)(B * S);
  switch (p.tr) {  // second argument = the most tiles assign_tiles gives one wave
    case 1: rc = launch_fused<1, 1>(ctx, p, grid, lds); break;
    case 2: rc = launch_fused<2, 2>(ctx, p, grid, lds); break;
    case 3: rc = launch_fused<3, 3>(ctx, p, grid, lds); break;
    case 4: rc = launch_fused<4, 2>(ctx, p, grid, lds); break;
    case 5: rc = launch_fused<5, 4>(ctx, p, grid, lds); break;
    case 6: rc = launch_fused<6, 6>(ctx, p, grid, lds); break;
    case 7: rc = launch_fused<7, 7>(ctx, p, grid, lds); break;
    default: rc = launch_fused<8, 8>(ctx, p, grid, lds); break;
  }
  if (rc != RT_OK) return rc;
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
