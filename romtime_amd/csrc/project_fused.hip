// Fused reduced projection  A_N[b] = V^T (A_b V)  for B value-vectors on one CSR pattern
// (project_csr, utils.py:96-113; MDEIM.project_basis, mdeim.py:153-192; the online direct path,
// rom.py:149-153) without materialising A_b V in HBM.
//
// One workgroup (8 waves) owns one (value-vector b, DoF-range) pair and walks the range in stages of 32 rows:
//   * the column window of the stage's rows (<= 48 DoFs for banded FE operators) is copied from V into LDS
//     once; the stage's own 32 rows inside it are the A operand of the product,
//   * the 32 rows of (A_b V) are built on the fly, row k = sum_e a_b[e] V[col_e][:], out of that window
//     (or gathered through L1/L2 when the window is too wide) into the LDS image of the B operand,
//   * the r x r product is accumulated on the FP64 matrix cores; every wave owns fixed rectangular blocks of
//     the ceil(r/16)^2 grid of 16x16 tiles (compile-time layout per wave), so r = 80 costs 25 tiles, not the
//     36 of a padded 96 x 96 tile, and a k-step of an ni x nj block costs ni + nj LDS reads.
// Partial r x r blocks of the DoF-ranges go to slabs and are summed in a fixed order.
// HBM traffic: the value vectors once (8 nnz B bytes) + V once; the unfused path moved 16 N r B more.
//
// What shapes the code: on gfx950 a wave's FP64 MFMA blocks the VALU of its SIMD for the 64 cycles it runs --
// measured with rt_bench_mfma_f64's co-issue probe, MFMAs of one wave and integer VALU work of another wave on
// the same SIMD take exactly the sum of their times.  Every VALU instruction therefore costs matrix-core time,
// and a producer/consumer split of the waves buys nothing (tried: slower).  So all strides and tile offsets are
// compile-time (LDS accesses use immediate offsets), window loads are 16-byte with per-thread offsets computed
// once, each staged entry carries its window-row byte offset, and the stage records make every global load of
// stage s+1 independent of any other load (they are all in flight while stage s computes).
#include "common.h"
#include "tile_layout.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

// Timing-only builds (ROMTIME_EXTRA_HIPFLAGS=-DROMTIME_PF_ABLATE python -m romtime_amd.build, then ROMTIME_PF_FLAGS =
// 8: no buffer-descriptor fetch, 16: no gather, 32: no MFMAs, 64: no global loads, 16384: time stamps around the
// MFMA phases) price the phases of a stage; their results are wrong by design, so the switches do not exist in the
// shipped library.  Measured (32 vectors, N = 1e5, r = 80, ms): all 0.89 = skeleton (barriers, commit, scalar code)
// 0.18 + MFMAs 0.55 (the matrix-core time of the flops at the clock the chip holds) + gather 0.14 + fetch 0.02.
// The two workgroups a CU holds do run half a stage apart (time stamps: MFMA phase 4300 of a 10400-cycle stage,
// the second workgroup's 0.57 of a period behind the first's; a deliberate skew, a rotation of the heavy SIMD and
// a 1024-thread workgroup of two groups sharing barriers one phase apart changed nothing or lost), but the vector
// instructions of the other phases queue behind the 64-cycle MFMAs of the neighbour, so those phases stretch to
// 6100 cycles and the matrix cores idle 30 % of the time.  Every vector instruction outside the MFMA phase costs
// latency as well as issue time; that is what the stage loads through buffer descriptors, the affine window copy
// and the chunked gather are for.
#ifdef ROMTIME_PF_ABLATE
#define PF_ABLATE(bit) ((p.flags & (bit)) != 0)
#else
#define PF_ABLATE(bit) false
#endif

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PK = 32;        // rows per stage
constexpr int PT = 512;       // threads (8 waves)
constexpr int RMAX = 128;
constexpr int WROWS = 48;     // V rows kept in LDS per stage (banded FE operators touch ~PK + 2*bandwidth)
constexpr int EMAX = PT;      // entries of one stage staged through LDS (one per thread)

// == 16 (mod 32) doubles: conflict-free ds_read_b64 of the MFMA operands (16 tr already is for odd tr)
__host__ __device__ constexpr bool separate_a(int tr) { return tr <= 5; }   // see project_wave
__host__ __device__ constexpr int stride_of(int tr) { return 16 * tr + ((tr & 1) ? 0 : 16); }

// Everything a stage needs to start its loads, in one wave-uniform 16-byte record (one scalar load, issued a
// whole stage ahead): first entry e0 of the stage's rows, first row `lo` of their column window, and packed
//   bits  0..9   ne    entries of the stage (<= EMAX)
//   bits 10..16  nrow  rows of the window (<= WROWS); 0 = do not window (too wide / too many entries)
//   bits 17..26  mr    most entries in one row of the stage
//   bit  27      uni   every row of the stage has exactly mr entries
struct StageRec {
  long e0;
  int lo;
  unsigned packed;
};
__host__ __device__ inline int rec_ne(unsigned pk) { return (int)(pk & 1023u); }
__host__ __device__ inline int rec_nrow(unsigned pk) { return (int)((pk >> 10) & 127u); }
__host__ __device__ inline int rec_mr(unsigned pk) { return (int)((pk >> 17) & 1023u); }
__host__ __device__ inline bool rec_uni(unsigned pk) { return ((pk >> 27) & 1u) != 0; }

struct Entry {  // staged entry: value and byte offset of its V row inside the LDS window
  double val;
  int rowoff;
  int pad;
};
constexpr int EPAD = 8;       // zero entries behind the staged ones (a gather chunk may read past the last entry)

// The table starts with a 16-byte header: word 0 != 0 when some stage of the pattern cannot be windowed.
constexpr size_t TABLE_HEADER = 16;

struct ProjParams {
  const long* indptr;
  const long* indices;
  const double* data;
  long d_es, d_bs;
  const double* V;
  long ldv;
  double* slab;        // [B][S][r*r]
  int N, k_per_split;
  int r, B, S;
  int flags;
  int xcd_map;          // 1: workgroups of an XCD take a contiguous run of the range-major work list (see the kernel)
  const StageRec* rec;  // [stages]; handed to the kernel as an argument of its own (see project_fused_kernel)
  const int* any_unwindowed;
};

// One wave per stage (a thread per stage walked its ~160 entries one load after the other: 31 us at N = 1e5).
__global__ __launch_bounds__(256) void project_stages_kernel(const long* __restrict__ indptr, const long* __restrict__ indices, long N,
                                                             StageRec* __restrict__ rec, int* __restrict__ any_unwindowed) {
  const int lane = threadIdx.x & 63;
  const long st = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  const long k0 = st * PK;
  if (k0 >= N) return;
  const long k1 = (k0 + PK < N) ? k0 + PK : N;
  int lo = (int)k0, hi = (int)(k1 - 1);
  const long e0 = indptr[k0], e1 = indptr[k1];
  for (long e = e0 + lane; e < e1; e += 64) {
    const int c = (int)indices[e];
    lo = min(lo, c);
    hi = max(hi, c);
  }
  long mr = 0, mn = 1L << 40;
  if (k0 + lane < k1) mr = mn = indptr[k0 + lane + 1] - indptr[k0 + lane];   // PK <= 64 rows: one per lane
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off));
    hi = max(hi, __shfl_xor(hi, off));
    mr = max(mr, __shfl_xor(mr, off));
    mn = min(mn, __shfl_xor(mn, off));
  }
  if (lane != 0) return;
  const bool ok = (hi - lo + 1 <= WROWS) && (e1 - e0 <= EMAX);
  StageRec o;
  o.e0 = e0;
  o.lo = ok ? lo : (int)k0;
  o.packed = ok ? ((unsigned)(e1 - e0) | ((unsigned)(hi - lo + 1) << 10) | ((unsigned)mr << 17) | ((mr == mn && k1 - k0 == PK ? 1u : 0u) << 27)) : 0u;
  rec[st] = o;
  if (!ok) atomicOr(any_unwindowed, 1);
}

template <int NI, int NJ>
__device__ __forceinline__ void block_store(double* out, int r, int i0, int j0, int l4, int l15,
                                            const d4 (&acc)[NI * NJ > 0 ? NI * NJ : 1]) {
  if constexpr (NI * NJ > 0) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int row = 16 * (i0 + i) + l4 + 4 * c, col = 16 * (j0 + j) + l15;
          if (row < r && col < r) out[(long)row * r + col] = acc[i * NJ + j][c];
        }
  }
}

#ifdef ROMTIME_PF_ABLATE
// per workgroup: HW_ID | XCC_ID << 32, then s_memtime at the head and tail of the MFMA phases of stages 10..17
// (ROMTIME_PF_FLAGS & 16384; read back with rt_pf_debug_dump, tools/probes/proj_phase.py)
__device__ unsigned long long g_dbg[2048 * 20];
#endif

// The whole stage loop for the wave that owns block W of the layout (compile-time tiles).  Software-pipelined:
// while stage s is computed out of LDS, the entries, the per-row entry ranges and the V-row window of stage
// s+1 are in flight into registers (their addresses come from the stage record fetched during stage s-1, so
// no load waits on another load); they are written to LDS after the MFMAs of stage s.
template <int TR, int W, bool MIXED>
__device__ __forceinline__ void project_wave(const ProjParams& p, const StageRec* __restrict__ rec, double* sm) {
  constexpr int STRIDE = stride_of(TR);
  constexpr int PAIRS = 8 * TR;                               // d2 pairs per window row
  constexpr int RPP = PT / PAIRS;                             // window rows per pass of the workgroup
  constexpr int NWP = (WROWS + RPP - 1) / RPP;                // passes = d2 loads per thread and stage
  constexpr Blk b0 = Layout<TR>::blk[W];
  constexpr int NI = b0.ni, NJ = b0.nj, NT = NI * NJ;
  constexpr int U = TR <= 2 ? 6 : (TR <= 4 ? 3 : 2);          // entries per gather chunk (2 U TR transient VGPRs)
  static_assert(U <= EPAD, "a chunk may overrun the staged entries by U - 1");
  double* sW = sm;                                            // [WROWS][STRIDE]  window of V rows
  // r <= 80: the stage's own rows are copied to sA, so that the next window can be committed while the MFMAs run -
  // two barriers per stage instead of three (r = 64: 0.69 -> 0.63 ms per 32 vectors, r = 40: 0.52 -> 0.49, r = 80 with
  // 120 vectors: 3.12 -> 3.01 ms; at r = 80 the copy goes one pair at a time, its registers would spill otherwise).
  // Beyond, one workgroup per CU anyway and the A operand is read out of the window.
  constexpr bool SEP_A = separate_a(TR);
  double* sA = sm + WROWS * STRIDE;                           // [PK][STRIDE]     the stage's own V rows (SEP_A)
  double* sB = sA + (SEP_A ? PK * STRIDE : 0);                // [PK][STRIDE]     rows of A_b V (B operand)
  Entry* sEnt = reinterpret_cast<Entry*>(sB + PK * STRIDE);   // [EMAX + EPAD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int l15 = lane & 15, l4 = lane >> 4;
  // Which (value vector b, DoF range s).  Workgroup i runs on XCD i % 8 (checked for the Gram kernel: counter
  // gram_off_xcd), and each XCD has its own L2: with b = i % B the B readers of a range's rows of V were spread over all
  // eight XCDs and V came in from the fabric up to eight times (764 MB per 32-vector launch against 192 algorithmic).
  // Instead XCD x takes the x-th eighth of the work list ordered by (s, b): the readers of a range sit on one XCD - two
  // at a boundary - next to each other in dispatch order, and share its rows through that XCD's L2.
  int b, s;
  if (p.xcd_map) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int total = p.B * p.S, per = (total + 7) >> 3;
    const int i = x * per + j;
    if (j >= per || i >= total) return;   // padding of the grid (whole workgroup, before any barrier)
    s = i / p.B;
    b = i - s * p.B;
  } else {
    b = blockIdx.x % p.B;
    s = blockIdx.x / p.B;
  }
  const int r = p.r;
  const int kbeg = s * p.k_per_split;                         // multiple of PK
  const int kend = (kbeg + p.k_per_split < p.N) ? kbeg + p.k_per_split : p.N;
  const double* dat = p.data + (long)b * p.d_bs;
  const bool vec2 = ((p.ldv & 1) == 0) && ((reinterpret_cast<size_t>(p.V) & 15) == 0) && ((r & 1) == 0) && !PF_ABLATE(8);

  d4 acc[NT > 0 ? NT : 1];
#pragma unroll
  for (int q = 0; q < (NT > 0 ? NT : 1); ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};

  // window copy: pass q of the workgroup covers rows RPP q .. RPP q + RPP - 1 of the window; this thread takes
  // columns 2 cp, 2 cp + 1 of row wrow + RPP q.  Its global and LDS offsets differ between passes by
  // wave-uniform / compile-time amounts, so two registers describe all NWP loads.
  const int wrow = (tid < RPP * PAIRS) ? tid / PAIRS : (1 << 20);
  const int cp = tid % PAIRS;
  const unsigned w_goff = (unsigned)(((tid / PAIRS) * (int)p.ldv + 2 * cp) * 8);  // bytes
  double* w_lds = sW + (tid / PAIRS) * STRIDE + 2 * cp;
  const unsigned pass32 = (unsigned)(RPP * (int)p.ldv * 8);   // bytes between passes of the window copy
  const int kk = tid >> 4, jg = tid & 15;  // gather mapping: row kk of the stage, columns jg + 16 c

  // registers of the stage in flight
  d2 wreg[NWP];
  double vreg = 0.0;
  int creg = 0, nb_lo = 0, nb_hi = 0;

  // every global address below is (wave-uniform base) + (32-bit per-thread offset)
  const unsigned ent_off = (unsigned)tid * (unsigned)p.d_es * 8u, idx_off = (unsigned)tid * 8u, row_off = (unsigned)kk * 8u;
  auto ld = [](const void* base, unsigned off) { return *reinterpret_cast<const double*>(static_cast<const char*>(base) + off); };
  auto ldi = [](const void* base, unsigned off) { return *reinterpret_cast<const int*>(static_cast<const char*>(base) + off); };
  // Full windowed stages (all but the ends of the matrix) load through buffer descriptors: the address is
  // descriptor base + per-thread 32-bit offset (a constant of the thread) + a scalar offset, and lanes past the
  // stage's entries are dropped by the range check - not one vector instruction per load.  Window rows past the
  // stage's last column and the idle threads' rows are read too (they exist: lo + NWP RPP < N) and never used.
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.V), 0, 0xffffffffu, 0x00020000);
  auto fetch = [&](int k0, const StageRec& rc) {  // issue the loads of stage k0 (no LDS access, no dependent load)
    const int rows_left = (kend - k0 < PK) ? kend - k0 : PK;
    const int nr = rec_nrow(rc.packed);
    if (vec2 && nr > 0 && rows_left == PK && rc.lo + NWP * RPP < p.N) {
      const unsigned so = (unsigned)rc.lo * (unsigned)p.ldv * 8u;
#pragma unroll
      for (int q = 0; q < NWP; ++q)
        wreg[q] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rsV, w_goff, so + q * pass32, 0));
      const unsigned ne = (unsigned)rec_ne(rc.packed);
      const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(dat + rc.e0 * p.d_es), 0, ne * (unsigned)p.d_es * 8u, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<long*>(p.indices + rc.e0), 0, ne * 8u, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<long*>(p.indptr + k0), 0, (PK + 1) * 8u, 0x00020000);
      vreg = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsD, ent_off, 0, 0));
      creg = (int)__builtin_amdgcn_raw_buffer_load_b32(rsI, idx_off, 0, 0);   // low word (< 2^31); `- lo` waits for commit()
      // two dword loads: a wider one leaves dead destination registers, which the allocator hands out again and
      // the hazard logic then protects with a full vmcnt(0) in the middle of the gather
      nb_lo = (int)__builtin_amdgcn_raw_buffer_load_b32(rsP, row_off, 0, 0);
      nb_hi = (int)__builtin_amdgcn_raw_buffer_load_b32(rsP, row_off + 8u, 0, 0);
      return;
    }
    const int nrow = nr > 0 ? nr : rows_left;
    const char* vbase = reinterpret_cast<const char*>(p.V + (long)rc.lo * p.ldv);  // wave-uniform
#pragma unroll
    for (int q = 0; q < NWP; ++q) {
      d2 v{0.0, 0.0};
      if (wrow < nrow - RPP * q) {
        const char* src = vbase + (w_goff + q * pass32);
        if (2 * cp < r) v.x = *reinterpret_cast<const double*>(src);
        if (2 * cp + 1 < r) v.y = *reinterpret_cast<const double*>(src + 8);
      }
      wreg[q] = v;
    }
    if (nr > 0) {
      const bool v = tid < rec_ne(rc.packed);
      vreg = v ? ld(dat + rc.e0 * p.d_es, ent_off) : 0.0;
      creg = v ? ldi(p.indices + rc.e0, idx_off) : rc.lo;
      const bool rv = kk < rows_left;
      nb_lo = rv ? ldi(p.indptr + k0, row_off) : 0;
      nb_hi = rv ? ldi(p.indptr + k0 + 1, row_off) : 0;
    }
  };
  int c_lo = 0, row_cnt = 0;
  unsigned c_pk = 0;   // packed fields of the committed stage
  const Entry* ep = sEnt;
  auto commit = [&](const StageRec& rc) {  // registers -> LDS; rc = the record fetch() was given
    if (RPP * PAIRS == PT || wrow < RPP) {   // threads past the last whole row of a pass hold nothing
#pragma unroll
      for (int q = 0; q < NWP; ++q)
        if (RPP * (q + 1) <= WROWS || wrow < WROWS - RPP * q) *reinterpret_cast<d2*>(w_lds + q * RPP * STRIDE) = wreg[q];
    }
    Entry en;
    en.val = vreg;
    en.rowoff = (creg - rc.lo) * STRIDE * (int)sizeof(double);   // no arithmetic on loaded values before this point
    en.pad = 0;
    sEnt[tid] = en;
    c_lo = rc.lo;
    c_pk = rc.packed;
    ep = sEnt + (nb_lo - (int)rc.e0);
    row_cnt = nb_hi - nb_lo;
  };

  const int st0 = kbeg / PK, st1 = (kend + PK - 1) / PK;
  StageRec r1{}, r2{};
  if (st0 < st1) {
    r1 = rec[st0];
    if (st0 + 1 < st1) r2 = rec[st0 + 1];
    fetch(kbeg, r1);
    commit(r1);
  }
  __syncthreads();

  const char* sWc = reinterpret_cast<const char*>(sW) + jg * sizeof(double);
  double* brow = sB + kk * STRIDE + jg;
  const double* lB = sB + l4 * STRIDE + l15 + 16 * b0.j0;
  const double* lA0 = (SEP_A ? sA : sW) + l4 * STRIDE + l15 + 16 * b0.i0;
  constexpr int NAP = (PK + RPP - 1) / RPP;                   // passes of the copy of the stage's own rows
  __builtin_amdgcn_s_setprio(3);
  for (int st = st0; st < st1; ++st) {
    const int k0 = st * PK;
    const bool more = st + 1 < st1;
    if (more) {
      r1 = r2;                                    // loaded one stage ago
      if (st + 2 < st1) r2 = rec[st + 2];       // for the next iteration
      if (!PF_ABLATE(64)) fetch(k0 + PK, r1);
    }
    if constexpr (SEP_A) {   // A operand: window rows k0 - lo .. k0 - lo + PK - 1, copied LDS -> LDS
      if (RPP * PAIRS == PT || wrow < RPP) {
        const double* src = w_lds + (k0 - c_lo) * STRIDE;
        double* dst = sA + (w_lds - sW);
        if constexpr (TR <= 4) {
          d2 t[NAP];
#pragma unroll
          for (int q = 0; q < NAP; ++q)
            if (RPP * (q + 1) <= PK || wrow < PK - RPP * q) t[q] = *reinterpret_cast<const d2*>(src + q * RPP * STRIDE);
#pragma unroll
          for (int q = 0; q < NAP; ++q)
            if (RPP * (q + 1) <= PK || wrow < PK - RPP * q) *reinterpret_cast<d2*>(dst + q * RPP * STRIDE) = t[q];
        } else {   // registers are scarce: one pair at a time
#pragma unroll
          for (int q = 0; q < NAP; ++q)
            if (RPP * (q + 1) <= PK || wrow < PK - RPP * q) {
              const d2 t = *reinterpret_cast<const d2*>(src + q * RPP * STRIDE);
              __builtin_amdgcn_sched_barrier(0);
              *reinterpret_cast<d2*>(dst + q * RPP * STRIDE) = t;
            }
        }
      }
    }
    // B operand: rows of A_b V.  The entries of a row are taken U at a time: U entry reads, then their U TR
    // window reads, then the FMAs - two LDS round trips per chunk (one read after another costs four per entry).
    {
      double o[TR];
#pragma unroll
      for (int c = 0; c < TR; ++c) o[c] = 0.0;
      if (PF_ABLATE(16)) {
      } else if (rec_nrow(c_pk) > 0) {
        auto chunk = [&](auto uc, auto masked, int q0) {
          constexpr int UC = decltype(uc)::value;
          Entry en[UC];
#pragma unroll
          for (int u = 0; u < UC; ++u) en[u] = ep[q0 + u];
          if constexpr (decltype(masked)::value) {
#pragma unroll
            for (int u = 0; u < UC; ++u) en[u].val = (q0 + u < row_cnt) ? en[u].val : 0.0;
          }
          double x[UC][TR];
#pragma unroll
          for (int u = 0; u < UC; ++u) {
            const double* vr = reinterpret_cast<const double*>(sWc + en[u].rowoff);
#pragma unroll
            for (int c = 0; c < TR; ++c) x[u][c] = vr[16 * c];
          }
#pragma unroll
          for (int u = 0; u < UC; ++u)
#pragma unroll
            for (int c = 0; c < TR; ++c) o[c] = fma(en[u].val, x[u][c], o[c]);
        };
        auto row_sum = [&](auto masked) {
          const int c_mr = rec_mr(c_pk);
          int q0 = 0;
          for (; q0 + U <= c_mr; q0 += U) chunk(std::integral_constant<int, U>{}, masked, q0);
          const int rest = c_mr - q0;   // wave-uniform
          if constexpr (U > 1) { if (rest == 1) chunk(std::integral_constant<int, 1>{}, masked, q0); }
          if constexpr (U > 2) { if (rest == 2) chunk(std::integral_constant<int, 2>{}, masked, q0); }
          if constexpr (U > 3) { if (rest == 3) chunk(std::integral_constant<int, 3>{}, masked, q0); }
          if constexpr (U > 4) { if (rest == 4) chunk(std::integral_constant<int, 4>{}, masked, q0); }
          if constexpr (U > 5) { if (rest == 5) chunk(std::integral_constant<int, 5>{}, masked, q0); }
        };
        // rows of equal length (the interior of a structured operator) need no per-thread predicate
        if (rec_uni(c_pk)) row_sum(std::false_type{});
        else row_sum(std::true_type{});
      } else if (MIXED && k0 + kk < kend) {   // a stage that could not be windowed: straight out of global memory
        const long k = k0 + kk;
        const long e0 = p.indptr[k], e1 = p.indptr[k + 1];
        for (long e = e0; e < e1; ++e) {
          const double a = dat[e * p.d_es];
          const double* vr = p.V + p.indices[e] * p.ldv;
#pragma unroll
          for (int c = 0; c < TR; ++c)
            if (jg + 16 * c < r) o[c] = fma(a, vr[jg + 16 * c], o[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < TR; ++c) brow[16 * c] = o[c];
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
#ifdef ROMTIME_PF_ABLATE
    if (W == 0 && (p.flags & 16384) && blockIdx.x < 2048 && st - st0 >= 10 && st - st0 < 18 && lane == 0)
      g_dbg[blockIdx.x * 20 + 2 + 2 * (st - st0 - 10)] = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (NT > 0) if (!PF_ABLATE(32)) {
      // operands of k-step k4 + 1 are read before the MFMAs of k-step k4 are issued
      const double* lA = SEP_A ? lA0 : lA0 + (k0 - c_lo) * STRIDE;  // LDS row of DoF k0 (+ this lane's k)
      double a[2][NI], bq[2][NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[0][i] = lA[16 * i];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bq[0][j] = lB[16 * j];
#pragma unroll
      for (int k4 = 0; k4 < PK / 4; ++k4) {
        const int cur = k4 & 1, nxt = cur ^ 1;
        if (k4 + 1 < PK / 4) {
#pragma unroll
          for (int i = 0; i < NI; ++i) a[nxt][i] = lA[(k4 + 1) * 4 * STRIDE + 16 * i];
#pragma unroll
          for (int j = 0; j < NJ; ++j) bq[nxt][j] = lB[(k4 + 1) * 4 * STRIDE + 16 * j];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs they are to hide behind
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i * NJ + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][i], bq[cur][j], acc[i * NJ + j], 0, 0, 0);
      }
    }
#ifdef ROMTIME_PF_ABLATE
    if (W == 0 && (p.flags & 16384) && blockIdx.x < 2048 && st - st0 >= 10 && st - st0 < 18 && lane == 0) {
      g_dbg[blockIdx.x * 20 + 3 + 2 * (st - st0 - 10)] = __builtin_amdgcn_s_memtime();
      g_dbg[blockIdx.x * 20] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 32);
    }
#endif
    if constexpr (SEP_A) {
      if (more) commit(r1);            // at the MFMA phase's priority: raising it first cost 5 % at r = 64
      __builtin_amdgcn_s_setprio(3);   // the gather and fetch phase wins the issue slot over the other workgroup's MFMAs
    } else {
      __syncthreads();                 // the window is the A operand: every wave's MFMAs first
      if (more) commit(r1);
      __builtin_amdgcn_s_setprio(3);   // the gather and fetch phase wins the issue slot over the other workgroup's MFMAs
    }
    __syncthreads();
  }
  double* out = p.slab + ((long)b * p.S + s) * ((long)r * r);
  block_store<NI, NJ>(out, r, b0.i0, b0.j0, l4, l15, acc);
}

// MIXED = false is the kernel for patterns whose stages are all windowed (banded operators): without the
// global-memory gather in the same code the hazard logic keeps the loads of stage s+1 in flight across the whole of
// stage s (with it, a vmcnt(0) lands in front of the MFMAs).  Both variants are launched; the one the table's
// header does not ask for returns at once, so the host never has to read the header back.
template <int TR, bool MIXED>
__global__ __launch_bounds__(PT, TR <= 5 ? 4 : 2) void project_fused_kernel(const ProjParams p, const StageRec* __restrict__ rec,
                                                                            const int* __restrict__ any_unwindowed) {
  // rec and the header are arguments of their own: only a __restrict__ kernel argument tells the compiler that
  // nothing in the kernel writes the table, and only then does it fetch the records with scalar loads (through
  // the struct it used a vector load + vmcnt(0) + readfirstlane at the head of every stage)
  if ((*any_unwindowed != 0) != MIXED) return;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  // a short last stage lets the A operand reach rows past the window (they only meet zero rows of B, but must be
  // finite): no LDS word is ever read uninitialised; the entries behind the staged ones stay zero
  constexpr int words = (WROWS + (separate_a(TR) ? 2 : 1) * PK) * stride_of(TR) + 2 * (EMAX + EPAD);
  for (int i = threadIdx.x; i < words; i += PT) sm[i] = 0.0;
  __syncthreads();
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {  // one code path per wave: its tiles are constants
    case 0: project_wave<TR, 0, MIXED>(p, rec, sm); break;
    case 1: project_wave<TR, 1, MIXED>(p, rec, sm); break;
    case 2: project_wave<TR, 2, MIXED>(p, rec, sm); break;
    case 3: project_wave<TR, 3, MIXED>(p, rec, sm); break;
    case 4: project_wave<TR, 4, MIXED>(p, rec, sm); break;
    case 5: project_wave<TR, 5, MIXED>(p, rec, sm); break;
    case 6: project_wave<TR, 6, MIXED>(p, rec, sm); break;
    default: project_wave<TR, 7, MIXED>(p, rec, sm); break;
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

size_t rt_project_stage_table_bytes(int64_t N) { return TABLE_HEADER + sizeof(StageRec) * (size_t)((N + PK - 1) / PK); }

int rt_project_stage_table_banded(rt_ctx* ctx, const void* table, int* banded) {
  int any = 0;
  RT_HIP_CHECK(ctx, hipMemcpyAsync(&any, table, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  *banded = any ? 0 : 1;
  return RT_OK;
}

int rt_project_stage_table(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, int64_t N, void* table) {
  const long stages = (N + PK - 1) / PK;
  RT_HIP_CHECK(ctx, hipMemsetAsync(table, 0, TABLE_HEADER, ctx->stream));
  hipLaunchKernelGGL(project_stages_kernel, dim3((unsigned)((stages + 3) / 4)), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const long*>(indptr), reinterpret_cast<const long*>(indices), (long)N,
                     reinterpret_cast<StageRec*>(static_cast<char*>(table) + TABLE_HEADER), static_cast<int*>(table));
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

template <int TR>
static constexpr size_t fused_lds() {
  return sizeof(double) * (size_t)(WROWS + (separate_a(TR) ? 2 : 1) * PK) * stride_of(TR) + sizeof(Entry) * (EMAX + EPAD);
}

// workgroups of project_fused_kernel<TR> that fit one CU (LDS and registers), asked of the runtime once per TR
template <int TR>
static int fused_blocks_per_cu(rt_ctx* ctx, int* out) {
  static int cached = 0;  // occupancy is a property of the kernel on gfx950, the same on every device of the node
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&project_fused_kernel<TR, false>), 136 * 1024));
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&project_fused_kernel<TR, true>), 136 * 1024));
  if (cached == 0) {
    int nb = 0;
    RT_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, project_fused_kernel<TR, false>, PT, fused_lds<TR>()));
    cached = nb > 0 ? nb : 1;
  }
  *out = cached;
  return RT_OK;
}

static int fused_blocks_per_cu(rt_ctx* ctx, int tr, int* out) {
  switch (tr) {
    case 1: return fused_blocks_per_cu<1>(ctx, out);
    case 2: return fused_blocks_per_cu<2>(ctx, out);
    case 3: return fused_blocks_per_cu<3>(ctx, out);
    case 4: return fused_blocks_per_cu<4>(ctx, out);
    case 5: return fused_blocks_per_cu<5>(ctx, out);
    case 6: return fused_blocks_per_cu<6>(ctx, out);
    case 7: return fused_blocks_per_cu<7>(ctx, out);
    default: return fused_blocks_per_cu<8>(ctx, out);
  }
}

template <int TR>
static int launch_fused(rt_ctx* ctx, const ProjParams& p, unsigned grid, int banded) {
  if (banded != 0)
    hipLaunchKernelGGL((project_fused_kernel<TR, false>), dim3(grid), dim3(PT), fused_lds<TR>(), ctx->stream, p, p.rec, p.any_unwindowed);
  if (banded != 1)
    hipLaunchKernelGGL((project_fused_kernel<TR, true>), dim3(grid), dim3(PT), fused_lds<TR>(), ctx->stream, p, p.rec, p.any_unwindowed);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

// Returns RT_ERR_UNSUPPORTED for r > 128 (the caller then uses the unfused path).  `stage_table` = the table of
// rt_project_stage_table for this pattern, or nullptr (built here, one small launch).
int rt_project_fused(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, const double* data, int64_t d_es,
                     int64_t d_bs, int64_t B, int64_t N, const double* V, int64_t ldv, int64_t r, double* AN,
                     const void* stage_table, int banded) {
  if (r > RMAX) return RT_ERR_UNSUPPORTED;
  if (((int64_t)WROWS * ldv + RMAX) * 8 >= (1LL << 31) || d_es * 8 * EMAX >= (1LL << 31) || N >= (1LL << 31) - 2 * PK ||
      N * ldv * 8 >= (1LL << 32))
    return RT_ERR_UNSUPPORTED;  // 32-bit per-thread byte offsets, 32-bit row counters
#ifdef ROMTIME_PF_ABLATE
  static const int pf_flags = [] { const char* e = getenv("ROMTIME_PF_FLAGS"); return e ? atoi(e) : 0; }();
#else
  const int pf_flags = 0;
#endif
  ProjParams p;
  p.indptr = reinterpret_cast<const long*>(indptr);
  p.indices = reinterpret_cast<const long*>(indices);
  p.data = data; p.d_es = d_es; p.d_bs = d_bs;
  p.V = V; p.ldv = ldv; p.N = (int)N; p.r = (int)r; p.B = (int)B; p.flags = pf_flags;
  const int tr = (int)((r + 15) / 16), rp = tr * 16;
  // DoF ranges per value vector: about two rounds of resident workgroups (so that gather and MFMA phases of
  // different workgroups overlap), chosen so that the LAST round is full too - 120 vectors x 9 ranges on 512
  // resident workgroups ran 2.1 rounds' worth of work in 3 rounds (C4's projection at 0.36 of the MFMA peak next
  // to C5's 0.51 with 32 x 32 = exactly two rounds); 120 x 8 = 960 fills 2 rounds to 94 %.
  int per_cu = 1;
  int rc0 = fused_blocks_per_cu(ctx, tr, &per_cu);
  if (rc0 != RT_OK) return rc0;
  const long resident = (long)per_cu * ctx->num_cus;
  const long stages = (N + PK - 1) / PK;
  // one to four rounds of resident workgroups: the fullest last round wins (120 vectors: 17 ranges = 3.98 rounds run
  // at 0.66 of the peak, 8 ranges = 1.9 rounds at 0.62), ties go to the count closest to two rounds (32 vectors:
  // 16, 32, 48 and 64 ranges all fill their rounds; 32 measured best)
  long s_hi = (4 * resident + B - 1) / B;
  if (s_hi > stages / 8) s_hi = stages / 8;   // at least 8 stages per workgroup
  if (s_hi < 1) s_hi = 1;
  long s_lo = resident / B;
  if (s_lo < 1) s_lo = 1;
  if (s_lo > s_hi) s_lo = s_hi;
  long S = s_hi, kps = 0;
  double best = -1.0, best_off = 1e30;
  for (long cand = s_lo; cand <= s_hi; ++cand) {
    const long per = ((N + cand - 1) / cand + PK - 1) / PK * PK;
    const long s_eff = (N + per - 1) / per;
    const long g = B * s_eff, rounds = (g + resident - 1) / resident;
    const double fill = (double)g / (double)(rounds * resident);
    const double off = fabs((double)g / (double)resident - 2.0);
    if (fill > best + 1e-9 || (fill > best - 1e-9 && off < best_off)) { best = fill; best_off = off; S = s_eff; kps = per; }
  }
  p.k_per_split = (int)kps;
  p.S = (int)S;
  void* slab = nullptr;
  const size_t slab_bytes = (sizeof(double) * (size_t)B * S * r * r + 255) / 256 * 256;
  int rc = rt_scratch(ctx, slab_bytes + (stage_table ? 0 : rt_project_stage_table_bytes(N)), &slab);
  if (rc != RT_OK) return rc;
  p.slab = static_cast<double*>(slab);
  if (stage_table) {
    p.any_unwindowed = static_cast<const int*>(stage_table);
    p.rec = reinterpret_cast<const StageRec*>(static_cast<const char*>(stage_table) + TABLE_HEADER);
  } else {
    void* table = static_cast<char*>(slab) + slab_bytes;
    rc = rt_project_stage_table(ctx, indptr, indices, N, table);
    if (rc != RT_OK) return rc;
    p.any_unwindowed = static_cast<const int*>(table);
    p.rec = reinterpret_cast<const StageRec*>(static_cast<const char*>(table) + TABLE_HEADER);
  }
  if (ctx->profile) {
    if (!ctx->ev0) {
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev0));
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev1));
    }
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  }
  static const int xcd_map = [] { const char* e = getenv("ROMTIME_PROJECT_MAP"); return e ? atoi(e) : 1; }();
  p.xcd_map = xcd_map;
  const unsigned grid = xcd_map ? (unsigned)(((B * S + 7) / 8) * 8) : (unsigned)(B * S);
  switch (tr) {
    case 1: rc = launch_fused<1>(ctx, p, grid, banded); break;
    case 2: rc = launch_fused<2>(ctx, p, grid, banded); break;
    case 3: rc = launch_fused<3>(ctx, p, grid, banded); break;
    case 4: rc = launch_fused<4>(ctx, p, grid, banded); break;
    case 5: rc = launch_fused<5>(ctx, p, grid, banded); break;
    case 6: rc = launch_fused<6>(ctx, p, grid, banded); break;
    case 7: rc = launch_fused<7>(ctx, p, grid, banded); break;
    default: rc = launch_fused<8>(ctx, p, grid, banded); break;
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

#ifdef ROMTIME_PF_ABLATE
extern "C" int rt_pf_debug_dump(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif
