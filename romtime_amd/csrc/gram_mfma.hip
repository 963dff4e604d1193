// Snapshot Gram matrix G = X^T X for n >= 97 columns and long X: the dominant kernel of the POD.
//
// Differences from the generic strided GEMM (gemm_mfma.hip), all aimed at the two losses its
// profile showed (profiles/r01_v1_*: 20 % of the MFMAs spent below the diagonal, L2 hit rate 26 %):
//
//  * XCD-local K ranges.  XCD x (workgroups with blockIdx % 8 == x) owns rows [x K/8, (x+1) K/8).
//    Every workgroup of that XCD sweeps the SAME range, taking 16-row stages q, q+S, q+2S, ... of
//    it, so at any moment the 64 workgroups of an XCD read a window of a few hundred consecutive
//    rows: each 128-column panel is fetched from HBM once and re-read from that XCD's L2 by the
//    other tiles.  A workgroup that runs ahead misses in L2 and slows down, one that lags hits and
//    catches up, so the window stays together without any inter-workgroup protocol.
//  * Diagonal tiles do only the 36 MFMA tiles on/above the diagonal (of 64), 9 per wave:
//    executed/algorithmic flops 0.80 -> 0.97 at n = 512.  They run as a second launch of a
//    DIAG instantiation (72 accumulator VGPRs instead of 128, A panel only), each launch with
//    its own number of sub-splits so that it fills the chip evenly.
//
//
//  * Round 3, an alternative kept behind ROMTIME_GRAM_FLAGS & 16 (it measured no faster, see rt_gram128): ONE launch.
//    The two launches each sweep all of X (8.2 GB of HBM reads at best, 13 GB measured in the pipeline) and each leave
//    slots of the chip idle (6 tiles do not divide 64 slots).  In the one-launch form every workgroup slot of an XCD
//    carries up to two SEGMENTS - (tile, stage stride and offset, stage
//    range of the XCD's K range) - run one after the other with the accumulators flushed to a slab in between:
//      - an off-diagonal tile gets S_off slots that take its stages q, q + S_off, ... over the whole range;
//      - a diagonal tile (36 of 64 MFMA tiles: cheaper per stage, but 6 S_off + 4 S_diag = slots has no solution with
//        equal time per slot) gets s_d dedicated slots for the first phi of the range, and its LAST (1 - phi) is dealt
//        to off-diagonal slots as their second segment - they reach the end of the K range at the same time as the
//        dedicated slots get there, so every panel is still read from HBM once per XCD and shared through the L2
//        by all tiles that need it, and all slots finish together (gram_plan()).
//
// Output: per-(XCD, segment) 128x128 slabs, summed in a fixed order by gram_reduce_kernel
// (bitwise reproducible; exactly symmetric G).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gemm_panel.h"

using namespace rtk;

namespace {

constexpr int BT = 128;
constexpr int GT = 512;         // threads per workgroup: 8 waves as 2 (M) x 4 (N), 64 x 32 per wave
constexpr int MAX_SLOTS = 64;   // workgroups per XCD
constexpr int MAX_TILES = 36;   // upper-triangular 128-tiles: n <= 1024
typedef unsigned u2 __attribute__((ext_vector_type(2)));
constexpr unsigned long long PACE_ONE = 1ull << 40;   // one member of an XCD's pack (high 24 bits of its word)
constexpr int PACE_DROP = 6;    // own stages behind the pack's mean beyond which a workgroup leaves the pack

struct GramParams {
  const double* X;
  long ks, ms;        // X(k, i) at X[k*ks + i*ms]
  long K, n, kx;      // rows, columns, rows per XCD range (multiple of KB)
  long last0;         // first column of the last tile row / column: (tiles1 - 1) * 128, or n - 128 (see rt_gram128)
  double* slab;       // [8][nslots][128*128]
  int nslots, tiles1, vec, flags;  // flags: 1 = s_setprio around the MFMA block, 2 = stagger odd wave slots
  long* counters;      // rt_ctx::dev_counters
  unsigned long long* pace;  // this launch's progress counters ([8 XCDs][16]: {stages done, workgroups started}), or nullptr
  int pace_every, pace_slack, pace_naps;   // check every so many stages; lead allowed (stages of its own); naps per check
  unsigned char slot_tm[MAX_SLOTS], slot_tn[MAX_SLOTS], slot_q[MAX_SLOTS], slot_S[MAX_SLOTS];
};

// One launch (gram128_merged_kernel): up to two segments per slot.  S == 0: no such segment.  Stage indices are local
// to the XCD's K range; a segment takes stages s0 + q, s0 + q + S, ... below s1.
struct GramSegs {
  unsigned char tm[2][MAX_SLOTS], tn[2][MAX_SLOTS], q[2][MAX_SLOTS], S[2][MAX_SLOTS];
  unsigned short slab[2][MAX_SLOTS];
  int s0[2][MAX_SLOTS], s1[2][MAX_SLOTS];
};

// MFMA-tile sets of a diagonal 128x128 tile (8x8 grid of 16x16 tiles, only i <= j: 36 tiles): waves 0-3 take five
// consecutive tiles of the row-wise list each, waves 4-7 four.  Waves w and w + 4 of a workgroup share a SIMD, so
// every SIMD issues 9 MFMAs per k-step (5,5,5,5,5,5,4,2 put 10 on two of them: 10 % of the kernel).
// The operands of each MFMA are read from LDS at run-time offsets (no register-array indexing),
// so all waves run the same instruction stream.
__device__ const unsigned char kDiagTi[8][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 1, 1}, {1, 1, 1, 1, 1}, {2, 2, 2, 2, 2}, {2, 3, 3, 3, 255}, {3, 3, 4, 4, 255}, {4, 4, 5, 5, 255}, {5, 6, 6, 7, 255}};
__device__ const unsigned char kDiagTj[8][5] = {{0, 1, 2, 3, 4}, {5, 6, 7, 1, 2}, {3, 4, 5, 6, 7}, {2, 3, 4, 5, 6}, {7, 3, 4, 5, 0}, {6, 7, 4, 5, 0}, {6, 7, 5, 6, 0}, {7, 6, 7, 7, 0}};

// One segment: the 128 x 128 tile (tm, tn) of X^T X over the stages s0 + q0, s0 + q0 + S, ... (< s1, < the end of XCD
// x's K range) of XCD x's rows, written to `out` (a 128 x 128 slab).  All arguments are wave-uniform (SGPRs).
template <bool KC, bool DIAG>
__device__ __forceinline__ void gram_segment(const GramParams& p, double* smem, int x, int tm, int tn, int q0, int S,
                                             int s0, int s1, double* out) {
  using P = Panel<BT, KC, GT>;
  double* sA0 = smem;
  double* sA1 = smem + P::LDS;
  double* sB0 = smem + 2 * P::LDS;
  double* sB1 = smem + 3 * P::LDS;

  // opaque to the optimiser: nothing derived from the thread index is carried from one segment of the merged kernel
  // into the next (the off-diagonal loop has no VGPR to spare: values kept alive across segments were spilled in it)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const long m0 = (tm == p.tiles1 - 1) ? p.last0 : (long)tm * BT, n0 = (tn == p.tiles1 - 1) ? p.last0 : (long)tn * BT;
  const long kend_x = ((long)x * p.kx + p.kx < p.K) ? (long)x * p.kx + p.kx : p.K;
  const int nst_x = (kend_x > (long)x * p.kx) ? (int)((kend_x - (long)x * p.kx + KB - 1) / KB) : 0;
  const int send = s1 < nst_x ? s1 : nst_x;
  q0 += s0;                                                            // first stage of the segment
  const long kbeg = (long)x * p.kx;
  const long kend = kend_x;
  const int nstages = (send > q0) ? (send - q0 + S - 1) / S : 0;      // stages q0, q0+S, ...

  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int l15 = lane & 15, l4 = lane >> 4;

  // Pacing (a hint for L2 reuse, never a condition for progress).  The tiles of an XCD share a panel stage through the
  // L2 only while they read it within the few microseconds it stays there, and nothing keeps workgroups together by
  // itself: unpaced, the off-diagonal launch of 1e6 x 512 fetched 10.4 GB where perfect sharing needs 4.1; paced, 4.15 GB
  // and 2 % less time (tools/probes/gram_pace_ab.sh).  One 64-bit word per XCD holds the PACK: the sum of its members'
  // positions (stage index in the XCD's K range, low 40 bits) and their number (high 24 bits).  Every pace_every stages
  // lane 0 of wave 0 adds the workgroup's advance with one returning atomic, issued behind the stage's panel loads:
  //   - more than pace_slack of its own stages AHEAD of the pack's mean: nap ~0.45 us and look again, pace_naps times at most;
  //   - more than PACE_DROP of its own stages BEHIND (started late: its CU was busy, or it is a second batch): it leaves
  //     the pack - takes its position and itself out of the word - and runs unpaced, so the pack never waits for it.
  // Workgroups that have not started are not in the word; finished ones stay in it and look like leaders.
  unsigned long long* pace = p.pace ? p.pace + 16 * x : nullptr;
  int pacer = (pace && __builtin_amdgcn_readfirstlane(wid) == 0 && nstages > 0) ? 1 : 0;
  int pace_pos = 0, pace_cnt = 0;
  if (pacer && lane == 0) __hip_atomic_fetch_add(pace, PACE_ONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  constexpr int NACC = DIAG ? 5 : 8;
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  int dti[5], dtj[5];  // panel-local column offsets (x16) of this wave's diagonal MFMA tiles
  bool dval[5];
  {
    const int w = __builtin_amdgcn_readfirstlane(wid);
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      dval[q] = kDiagTi[w][q] != 255;
      dti[q] = dval[q] ? 16 * kDiagTi[w][q] : 0;
      dtj[q] = 16 * kDiagTj[w][q];
    }
  }

  d2 ra[P::NL], rb[DIAG ? 1 : P::NL];
  // interior tiles take the predicate-free loader for every stage that lies fully inside the K range
  const bool interior = p.vec && (m0 + BT <= p.n) && (n0 + BT <= p.n);
  if (p.flags & 2) {
    // de-phase the two waves that share a SIMD: the odd hardware wave slot starts half a stage late
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((4 - 1) << 11));  // HW_REG_HW_ID[3:0] = wave_id
    if (hw & 1) __builtin_amdgcn_s_sleep(DIAG ? 9 : 16);
  }
  if (nstages > 0) {
    const long k0 = kbeg + (long)q0 * KB;
    P::load(ra, p.X, p.ks, p.ms, k0, kend, m0, p.n, p.vec, tid);
    if constexpr (!DIAG) P::load(rb, p.X, p.ks, p.ms, k0, kend, n0, p.n, p.vec, tid);
    P::store(ra, sA0, tid);
    if constexpr (!DIAG) P::store(rb, sB0, tid);
  }
  __syncthreads();

  // scalar-base addressing of the refills (interior tiles): uniform pointers to the panel rows of
  // the NEXT stage, advanced by scalar adds; one per-thread byte offset for every load of the kernel
  const long k1 = kbeg + (long)(q0 + S) * KB;   // first row of stage 1
  const char* gA = reinterpret_cast<const char*>(p.X + k1 * p.ks + m0 * p.ms);
  const char* gB = reinterpret_cast<const char*>(p.X + k1 * p.ks + n0 * p.ms);
  const unsigned voff = P::lane_offset(p.ks, p.ms, tid);
  const long rows_step = P::load_step(p.ks, p.ms), stage_step = (long)S * KB * p.ks * 8;

  // stages whose 16 rows lie fully inside the K range: all of them, or all but the last (32-bit scalar compares in the loop)
  int n_full = nstages;
  if (nstages > 0 && kbeg + (long)(q0 + (nstages - 1) * S) * KB + KB > kend) n_full = nstages - 1;
  if (!interior) n_full = 0;

  auto compute = [&](const double* cA, const double* cB) {
    if constexpr (DIAG) {
#pragma unroll
      for (int k4 = 0; k4 < KB / 4; ++k4) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          if (!dval[q]) continue;  // wave-uniform
          const double a = P::frag(cA, dti[q], k4, l15, l4);
          const double b = P::frag(cA, dtj[q], k4, l15, l4);
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int k4 = 0; k4 < KB / 4; ++k4) {
        double a[4], b[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = P::frag(cA, wm * 64 + i * 16, k4, l15, l4);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = P::frag(cB, wn * 32 + j * 16, k4, l15, l4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
      }
    }
  };

  int st = 0;
  {
    // Fast loop (interior tile, either memory order): every stage here refills the other buffer from a stage that lies
    // fully inside the K range, through the scalar-base loader; unrolled by two so that the buffer parity is a
    // compile-time constant and buffer selection folds into the immediate offsets of the ds instructions.  Nothing of
    // the general (predicated) loader is live in it.
    auto fast_stage = [&](auto parity) {
      constexpr int PAR = decltype(parity)::value;
      P::load_full_u(ra, gA, voff, rows_step);
      if constexpr (!DIAG) P::load_full_u(rb, gB, voff, rows_step);
      // pacing: the advance goes in behind the panel loads and its answer (the word before the add) is back, like
      // them, by the time the stage's vmcnt wait is over - nothing waits for the L2 round trip
      const bool check = pacer && ++pace_cnt == p.pace_every;
      unsigned long long before = 0;
      if (check) {
        pace_cnt = 0;
        const int adv = S * p.pace_every;
        if (lane == 0) before = __hip_atomic_fetch_add(pace, (unsigned long long)adv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pace_pos += adv;
      }
      if (p.flags & 1) __builtin_amdgcn_s_setprio(1);
      compute(PAR ? sA1 : sA0, PAR ? sB1 : sB0);
      if (p.flags & 1) __builtin_amdgcn_s_setprio(0);
      gA += stage_step;
      gB += stage_step;
      P::store(ra, PAR ? sA0 : sA1, tid);
      if constexpr (!DIAG) P::store(rb, PAR ? sB0 : sB1, tid);
      if (check) {
        unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)before), hi = __builtin_amdgcn_readfirstlane((unsigned)(before >> 32));
        unsigned long long word = (((unsigned long long)hi << 32) | lo) + (unsigned long long)(S * p.pace_every);
#pragma unroll 1
        for (int nap = 0;; ++nap) {
          const long long total = (long long)(word & (PACE_ONE - 1)), members = (long long)(word >> 40);
          const long long lead = (long long)pace_pos * members - total;          // (mine - mean) x members
          if (lead < -(long long)PACE_DROP * S * members) {                     // a straggler: leave the pack
            if (lane == 0)
              __hip_atomic_fetch_add(pace, 0ull - (PACE_ONE + (unsigned long long)pace_pos), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
            pacer = 0;
            break;
          }
          if (lead <= (long long)p.pace_slack * S * members || nap >= p.pace_naps) break;
          // The window is tight (a panel stage lives in the L2 for about two rounds), so the control is firm: nap
          // ~0.45 us and look again, up to pace_naps times.  Gentler rules were measured and lose the sharing:
          // two short naps per stage of excess lead without polling fetched 9 GB, a slack of 4 stages 8 GB.
          __builtin_amdgcn_s_sleep(16);
          u2 v;  // the word straight from the L2 (scalar load past the scalar cache)
          asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(pace) : "memory");
          word = ((unsigned long long)v.y << 32) | v.x;
        }
      }
      __syncthreads();
    };
    if (!(p.flags & 4))
      for (; st + 2 < n_full; st += 2) {
        fast_stage(std::integral_constant<int, 0>{});
        fast_stage(std::integral_constant<int, 1>{});
      }
  }
  for (; st < nstages; ++st) {  // the remaining stages (all of them for column-major snapshots and edge tiles)
    const double* cA = (st & 1) ? sA1 : sA0;
    const double* cB = (st & 1) ? sB1 : sB0;
    const bool more = (st + 1 < nstages) && !(p.flags & 4);  // flags & 4: timing experiment, no refills
    if (more) {
      const long k0 = kbeg + (long)(q0 + (st + 1) * S) * KB;
      if (st + 1 < n_full) {
        P::load_full(ra, p.X, p.ks, p.ms, k0, m0, tid);
        if constexpr (!DIAG) P::load_full(rb, p.X, p.ks, p.ms, k0, n0, tid);
      } else {
        P::load(ra, p.X, p.ks, p.ms, k0, kend, m0, p.n, p.vec, tid);
        if constexpr (!DIAG) P::load(rb, p.X, p.ks, p.ms, k0, kend, n0, p.n, p.vec, tid);
      }
    }
    if (p.flags & 1) __builtin_amdgcn_s_setprio(1);
    compute(cA, cB);
    if (p.flags & 1) __builtin_amdgcn_s_setprio(0);
    if (more) {
      P::store(ra, (st & 1) ? sA0 : sA1, tid);
      if constexpr (!DIAG) P::store(rb, (st & 1) ? sB0 : sB1, tid);
    }
    __syncthreads();
  }

  if constexpr (DIAG) {
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (dval[q]) out[(dti[q] + l4 + 4 * r) * BT + dtj[q] + l15] = acc[q][r];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[(wm * 64 + i * 16 + l4 + 4 * r) * BT + wn * 32 + j * 16 + l15] = acc[i * 2 + j][r];
  }
}

// the XCD-local K ranges rest on workgroup i running on XCD i % 8: count the ones that do not (rt_ctx_get_counter
// "gram_off_xcd"; a CU-masked stream or a driver change could break the rule, the L2 reuse would go with it)
__device__ __forceinline__ void gram_check_xcd(const GramParams& p, int x) {
  if (threadIdx.x == 0 && (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) != x)
    atomicAdd(reinterpret_cast<unsigned long long*>(&p.counters[RT_CNT_GRAM_OFF_XCD]), 1ull);
}

// Two launches (short snapshot sets, where the plan of the merged kernel does not apply): one tile kind per launch.
template <bool KC, bool DIAG>
__global__ __launch_bounds__(GT, 4) void gram128_kernel(const GramParams p) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
  gram_check_xcd(p, x);
  // (a byte table in the kernel arguments indexed by a run-time slot is fetched with VECTOR loads: without the
  // readfirstlane every quantity derived from these four - tile origin, K range, panel pointers - lives in VGPRs and
  // all the "uniform" arithmetic of the loop is VALU work, which FP64 MFMAs cannot overlap with)
  const int tm = __builtin_amdgcn_readfirstlane((int)p.slot_tm[slot]), tn = __builtin_amdgcn_readfirstlane((int)p.slot_tn[slot]),
            q0 = __builtin_amdgcn_readfirstlane((int)p.slot_q[slot]), S = __builtin_amdgcn_readfirstlane((int)p.slot_S[slot]);
  gram_segment<KC, DIAG>(p, smem, x, tm, tn, q0, S, 0, 0x7fffffff, p.slab + ((long)x * p.nslots + slot) * (BT * BT));
}

// One launch: every slot runs its (up to) two segments, off-diagonal or diagonal as the plan says (gram_plan()).
template <bool KC>
__global__ __launch_bounds__(GT, 4) void gram128_merged_kernel(const GramParams p, const GramSegs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
  gram_check_xcd(p, x);
#pragma unroll 1
  for (int sg = 0; sg < 2; ++sg) {
    const int S = __builtin_amdgcn_readfirstlane((int)g.S[sg][slot]);
    if (S == 0) continue;
    const int tm = __builtin_amdgcn_readfirstlane((int)g.tm[sg][slot]), tn = __builtin_amdgcn_readfirstlane((int)g.tn[sg][slot]),
              q0 = __builtin_amdgcn_readfirstlane((int)g.q[sg][slot]), s0 = __builtin_amdgcn_readfirstlane(g.s0[sg][slot]),
              s1 = __builtin_amdgcn_readfirstlane(g.s1[sg][slot]), slab = __builtin_amdgcn_readfirstlane((int)g.slab[sg][slot]);
    double* out = p.slab + ((long)x * p.nslots + slab) * (BT * BT);
    if (tm == tn)
      gram_segment<KC, true>(p, smem, x, tm, tn, q0, S, s0, s1, out);
    else
      gram_segment<KC, false>(p, smem, x, tm, tn, q0, S, s0, s1, out);
    __syncthreads();   // the next segment reuses the LDS panels
  }
}

struct GramReduceParams {
  const double* slab_off;   // [8][nslots_off][128*128]
  const double* slab_diag;  // [8][nslots_diag][128*128]
  double* G;
  long n, last0;
  int nslots_off, nslots_diag, tiles1;
  unsigned char first[MAX_TILES], count[MAX_TILES];  // per upper-triangular tile, in its own launch's slots
  unsigned long long* pace;  // rt_ctx::gram_pace (256 words), zeroed here for the next Gram; or nullptr
};

// Each thread on or above the diagonal sums its element over the 8 XCD slabs x sub-splits in a fixed order (rows
// of a slab are contiguous, so the reads coalesce) and writes it to both (i, j) and (j, i): G is exactly symmetric
// and the slabs are read once (the mirrored half used to re-read them column-wise).
__global__ void gram_reduce_kernel(const GramReduceParams p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p.pace && idx < 256) p.pace[idx] = 0ull;
  if (idx >= p.n * p.n) return;
  const long i = idx / p.n, j = idx % p.n;
  if (i > j) return;
  const int tm = (int)(i / BT), tn = (int)(j / BT);
  const int t = tm * p.tiles1 - tm * (tm - 1) / 2 + (tn - tm);
  const long off = (i - (tm == p.tiles1 - 1 ? p.last0 : (long)tm * BT)) * BT + (j - (tn == p.tiles1 - 1 ? p.last0 : (long)tn * BT));
  const double* slab = (tm == tn) ? p.slab_diag : p.slab_off;
  const int nslots = (tm == tn) ? p.nslots_diag : p.nslots_off;
  const int cnt = p.count[t];
  const double* base = slab + (long)p.first[t] * (BT * BT) + off;
  double sum = 0.0;
  for (int x = 0; x < 8; ++x) {
    const double* src = base + (long)x * nslots * (BT * BT);
#pragma unroll 4
    for (int q = 0; q < cnt; ++q) sum += src[(long)q * (BT * BT)];
  }
  p.G[idx] = sum;
  if (i != j) p.G[j * p.n + i] = sum;
}

template <bool KC, bool DIAG>
int launch_gram(rt_ctx* ctx, const GramParams& p, int grid) {
  size_t lds = sizeof(double) * (DIAG ? 2 : 4) * Panel<BT, KC, GT>::LDS;
  if (p.flags & 128) lds = 100 * 1024;   // experiment: one workgroup per CU
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&gram128_kernel<KC, DIAG>), (int)lds));
  hipLaunchKernelGGL((gram128_kernel<KC, DIAG>), dim3(grid), dim3(GT), lds, ctx->stream, p);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

// The plan of the one-launch kernel: P slots per XCD, tiles1 x tiles1 tile grid (upper triangle), nst stages in an
// XCD's K range, rho = cost of a diagonal-tile stage relative to an off-diagonal one (36 of 64 MFMA tiles, one panel
// instead of two: 0.60 measured from the two-launch kernels' rates).  Equal time per slot:
//   T* = (n_off + rho n_d) / P  per stage of the range;  S_off = ceil(1 / T*) slots per off-diagonal tile leave each of
//   them T* - 1/S_off to spare; the other slots are dealt to the diagonal tiles (s_d each, differing by at most one);
//   a diagonal tile that needs more than s_d T* gets H off-diagonal slots as helpers for the last 1 - phi of the range,
//   phi from  phi rho / s_d = 1 / S_off + (1 - phi) rho / H  (dedicated slots and helpers finish together).
// Fills the segments, the slab index of every segment and the per-tile slab lists of the reduction.
// Slots per off-diagonal (a) and per diagonal tile (b) of the uniform plan below; returns the slowest slot's time per
// stage of the range, max(1/a, rho/b), or 0 when there is no such plan.
// `cap`: most slots a tile may get (short sets: at least 48 stages per workgroup).
double gram_uniform_ab(int P, int n_off, int n_d, double rho, int cap, int* a_out, int* b_out) {
  double best = 1e30;
  int best_slots = 0;
  *a_out = *b_out = 0;
  for (int a = 1; a <= cap && n_off * a + n_d <= P; ++a)
    for (int b = 1; b <= cap && n_off * a + n_d * b <= P; ++b) {
      const double T = (1.0 / a > rho / b) ? 1.0 / a : rho / b;
      const int used = n_off * a + n_d * b;
      if (T < best - 1e-12 || (T < best + 1e-12 && used < best_slots)) { best = T; *a_out = a; *b_out = b; best_slots = used; }
    }
  return *a_out ? best : 0.0;
}

// `uniform`: no helpers - every off-diagonal tile gets a slots and every diagonal tile b, (a, b) minimising the slowest
// slot's time max(1/a, rho/b) within P slots (fewest slots among equals).  All slots then move through the K range at
// about one rate (a vs b/rho stages per unit of time; pacing trims the rest), so ALL tiles of an XCD read a panel stage
// while it is in the L2: one read of X per Gram, at the price of slots that idle part of the time.
bool gram_plan(int P, int tiles1, int nst, double rho, GramSegs& g, unsigned char* first, unsigned char* count, int* nslabs,
               int* s_off_out, bool uniform = false) {
  const int n_d = tiles1, n_off = tiles1 * (tiles1 - 1) / 2;
  if (n_off < 1 || P > MAX_SLOTS) return false;
  const double W = n_off + rho * n_d, Tstar = W / P;
  int S_off = (int)(1.0 / Tstar - 1e-9) + 1;
  while (S_off >= 1 && n_off * S_off + n_d > P) --S_off;
  int uni_b = 0;
  if (uniform) {
    int best_a = 0;
    if (gram_uniform_ab(P, n_off, n_d, rho, nst / 48, &best_a, &uni_b) <= 0.0) return false;
    S_off = best_a;
  }
  if (S_off < 1 || nst < 48 * S_off) return false;   // short sets: the two-launch form with its own caps
  const int rem = P - n_off * S_off;
  int s_d[8], H[8] = {0}, n1[8];
  double need[8], total_need = 0.0;
  const double spare = Tstar - 1.0 / S_off;
  for (int i = 0; i < n_d; ++i) {
    s_d[i] = rem / n_d + (i < rem % n_d ? 1 : 0);
    need[i] = rho - s_d[i] * Tstar;
    if (need[i] < 1e-3 * rho || spare <= 1e-3 * Tstar) need[i] = 0.0;
    if (uniform) { s_d[i] = uni_b; need[i] = 0.0; }
    total_need += need[i];
  }
  int helpers_left = n_off * S_off;
  for (int i = 0; i < n_d; ++i) {
    n1[i] = 0x7fffffff;
    if (need[i] <= 0.0) continue;
    int h = (int)(n_off * S_off * need[i] / total_need);
    if (h > helpers_left) h = helpers_left;
    if (h > 255) h = 255;
    if (h < 1) continue;
    H[i] = h;
    helpers_left -= h;
    double phi = (1.0 / S_off + rho / h) / (rho / s_d[i] + rho / h);
    if (phi > 1.0) phi = 1.0;
    n1[i] = (int)(phi * nst + 0.5);
    if (n1[i] >= nst) { n1[i] = 0x7fffffff; helpers_left += h; H[i] = 0; }
  }
  for (int sg = 0; sg < 2; ++sg)
    for (int q = 0; q < MAX_SLOTS; ++q) {
      g.tm[sg][q] = g.tn[sg][q] = g.q[sg][q] = g.S[sg][q] = 0;
      g.slab[sg][q] = 0;
      g.s0[sg][q] = 0;
      g.s1[sg][q] = 0x7fffffff;
    }
  // off-diagonal tiles: slot = slab = tile-major index
  int slot = 0, slab = 0, t = 0, off_index[MAX_TILES];
  for (int a = 0; a < tiles1; ++a)
    for (int b = a; b < tiles1; ++b, ++t) {
      if (a == b) continue;
      first[t] = (unsigned char)slab;
      count[t] = (unsigned char)S_off;
      for (int q = 0; q < S_off; ++q, ++slot, ++slab) {
        g.tm[0][slot] = (unsigned char)a; g.tn[0][slot] = (unsigned char)b;
        g.q[0][slot] = (unsigned char)q; g.S[0][slot] = (unsigned char)S_off;
        g.slab[0][slot] = (unsigned short)slab;
      }
    }
  (void)off_index;
  // diagonal tiles: dedicated slots, then helpers (second segment of off-diagonal slots, taken in slot order so that
  // the helpers of one diagonal tile are spread over several off-diagonal tiles)
  int next_helper = 0;
  t = 0;
  for (int a = 0; a < tiles1; ++a)
    for (int b = a; b < tiles1; ++b, ++t) {
      if (a != b) continue;
      first[t] = (unsigned char)slab;
      count[t] = (unsigned char)(s_d[a] + H[a]);
      for (int q = 0; q < s_d[a]; ++q, ++slot, ++slab) {
        g.tm[0][slot] = g.tn[0][slot] = (unsigned char)a;
        g.q[0][slot] = (unsigned char)q; g.S[0][slot] = (unsigned char)s_d[a];
        g.s1[0][slot] = n1[a];
        g.slab[0][slot] = (unsigned short)slab;
      }
      for (int h = 0; h < H[a]; ++h, ++slab) {
        // helper slots: stride n_d through the off-diagonal slots, so consecutive helpers of a tile belong to different tiles
        const int hs = next_helper++;
        g.tm[1][hs] = g.tn[1][hs] = (unsigned char)a;
        g.q[1][hs] = (unsigned char)h; g.S[1][hs] = (unsigned char)H[a];
        g.s0[1][hs] = n1[a];
        g.slab[1][hs] = (unsigned short)slab;
      }
    }
  if (slab > 255) return false;
  *nslabs = slab;
  *s_off_out = S_off;
  return uniform ? slot <= P : slot == P;
}

// Which form the snapshot Gram takes for a K x n set on `num_cus` CUs - pure host logic, no GPU (rt_gram_plan_info lets
// the CPU-side tests check it).  form 0: not this kernel (the generic symmetric GEMM), 1: two launches (S_off / S_diag
// sub-splits per off-diagonal / diagonal tile and XCD), 2: one launch with uniform slots (a / b per tile).
struct GramChoice {
  int form, a, b, s_off, s_diag;
};
GramChoice gram_choose(int num_cus, long K, long n, double rho, int slots_per_cu = 2) {
  GramChoice c{0, 0, 0, 0, 0};
  const int tiles1 = (int)((n + BT - 1) / BT), ntiles = tiles1 * (tiles1 + 1) / 2, n_off = ntiles - tiles1;
  const int slots_max = slots_per_cu * num_cus / 8;
  if (n < 97 || ntiles > MAX_TILES || n_off > slots_max || slots_max > MAX_SLOTS || K < 8L * 64 * KB) return c;
  const long kx = ((K + 7) / 8 + KB - 1) / KB * KB;
  const int cap = (int)(kx / KB / 48);   // no tile gets more slots than leave 48 stages per workgroup
  if (cap < 1) return c;
  c.s_off = n_off ? std::min(slots_max / n_off, cap) : 0;
  c.s_diag = std::min(slots_max / tiles1, cap);
  if (n_off >= 1) {
    // One launch with uniform slots reads X once but leaves slots idle part of the time; taken when the slot model says
    // it costs at most 5 % more than two launches (measured 1-17 % faster there, profiles/r03_gram_pace_ab.txt), with at
    // least 16 workgroups per XCD, and not for two-tile-column sets while the cap binds (1e5 x 256: 0.219 vs 0.209 ms)
    const double t_uni = gram_uniform_ab(slots_max, n_off, tiles1, rho, cap, &c.a, &c.b);
    const double t_two = (c.s_off >= 1 && c.s_diag >= 1) ? 1.0 / c.s_off + rho / c.s_diag : 0.0;
    if (t_uni > 0.0 && t_two > 0.0 && t_uni <= 1.05 * t_two && n_off * c.a + tiles1 * c.b >= 16 && !(tiles1 == 2 && c.a >= cap)) {
      c.form = 2;
      return c;
    }
  }
  const int busiest = n_off ? n_off * c.s_off : tiles1 * c.s_diag;
  c.form = (busiest >= 16) ? 1 : 0;
  return c;
}

template <bool KC>
int launch_gram_merged(rt_ctx* ctx, const GramParams& p, const GramSegs& g, int grid) {
  constexpr size_t lds = sizeof(double) * 4 * Panel<BT, KC, GT>::LDS;
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&gram128_merged_kernel<KC>), (int)lds));
  hipLaunchKernelGGL((gram128_merged_kernel<KC>), dim3(grid), dim3(GT), lds, ctx->stream, p, g);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

}  // namespace

// Returns RT_ERR_UNSUPPORTED when the shape is outside this kernel's regime (caller falls back to
// the generic symmetric GEMM).
int rt_gram128(rt_ctx* ctx, const double* X, int64_t ks, int64_t ms, int64_t K, int64_t n, double* G) {
  const int tiles1 = (int)((n + BT - 1) / BT);
  const int ntiles = tiles1 * (tiles1 + 1) / 2;
  const int n_off = ntiles - tiles1;
  static const int occ1 = [] { const char* e = getenv("ROMTIME_GRAM_FLAGS"); return e ? (atoi(e) & 128) : 0; }();
  const int slots_max = (occ1 ? 1 : 2) * ctx->num_cus / 8;  // 2 workgroups per CU, per XCD
  if (n < 97 || ntiles > MAX_TILES || n_off > slots_max || slots_max > MAX_SLOTS || K < 8L * 64 * KB)
    return RT_ERR_UNSUPPORTED;
  if (!(ks == 1 || ms == 1)) return RT_ERR_UNSUPPORTED;
  const bool kc = (ks == 1) && (ms != 1);

  GramParams p;
  p.X = X; p.ks = ks; p.ms = ms; p.K = K; p.n = n;
  p.kx = ((K + 7) / 8 + KB - 1) / KB * KB;
  p.tiles1 = tiles1;
  const long other = kc ? ms : ks;
  p.vec = ((((uintptr_t)X) & 15) == 0 && (other % 2 == 0)) ? 1 : 0;
  // A last tile row / column that is not full (n no multiple of 128) would take the predicated loader for every stage:
  // 7.0 ms at n = 500 against 4.4 at n = 512.  Instead the last panel is SHIFTED to end at column n (it overlaps its
  // left neighbour; 16-byte alignment of its rows asks for an even n in row-major order): all panels are full, every tile
  // takes the fast loader, the entries computed twice are taken from the tile that owns them by the reduction.
  p.last0 = (long)(tiles1 - 1) * BT;
  if (n % BT != 0 && n >= BT && p.vec && (kc || n % 2 == 0)) p.last0 = n - BT;
  // the scalar-base loader keeps a 32-bit per-thread byte offset of up to 8 rows: beyond that, the predicated loader
  if ((!kc && (long)ks * 8 * 8 >= (1L << 31)) || (kc && (long)ms * 64 * 8 >= (1L << 31))) p.vec = 0;
  static const int env_flags = [] { const char* e = getenv("ROMTIME_GRAM_FLAGS"); return e ? atoi(e) : 1; }();
  p.flags = env_flags;
  p.counters = ctx->dev_counters;
  p.pace = nullptr;
  // pacing: every*100 + slack*10 + naps.  Two launches (equal workgroups, only jitter to correct): every 2 stages, 2
  // stages of slack; one launch (diagonal slots are ~1/6 faster and must be held back all the time): 1 stage of slack
  static const int pace_env = [] { const char* e = getenv("ROMTIME_GRAM_PACE"); return e ? atoi(e) : 0; }();
  auto set_pace = [&](int cfg) {
    if (pace_env) cfg = pace_env;
    p.pace_every = cfg / 100 > 0 ? cfg / 100 : 1; p.pace_slack = (cfg / 10) % 10; p.pace_naps = cfg % 10;
  };
  set_pace(228);
  if (ctx->gram_pace_on && !(env_flags & 32)) {
    if (!ctx->gram_pace) {
      RT_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->gram_pace), sizeof(unsigned long long) * 256));
      RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->gram_pace, 0, sizeof(unsigned long long) * 256, ctx->stream));
    }
    p.pace = ctx->gram_pace;
  }

  // ---- one launch (ROMTIME_GRAM_FLAGS & 16; long snapshot sets with off-diagonal tiles) --------------------------------
  // Measured (tools/probes/gram_merged_ab.sh, profiles/r03_gram_merged_ab.txt): perfectly balanced slots and one launch
  // buy nothing - 4.36 vs 4.34 ms back to back on the whole chip, 4.80 vs 4.66 ms inside the POD pipeline (224 CUs), HBM
  // reads 12.9 vs 13.6 GB.  The kernel runs at the board's power limit: the slots the two-launch form leaves idle (4 of
  // 64 in the off-diagonal launch) are not lost time, the busy ones clock higher.  Kept as a switch, not the default.
  static const int rho_pct = [] { const char* e = getenv("ROMTIME_GRAM_RHO"); return e ? atoi(e) : 60; }();
  // ---- which plan --------------------------------------------------------------------------------------------------------
  // One launch with uniform slots reads X once (4.15 GB on 1e6 x 512 against 8.2 for two paced launches and 14.5 unpaced)
  // but leaves slots idle part of the time; it is taken when the model says it costs at most 5 % more
  // than two launches - where it then measured 1-6 % FASTER (n = 256, 384, 512, 1024: tools/probes/gram_shapes.py);
  // n = 640 / 768 (model 1.15 / 1.07, measured 1.12 / 1.05) and the pipeline's 56 slots (1.08, measured 1.08) stay with
  // two launches.  ROMTIME_GRAM_FLAGS & 256 forces it, & 512 forbids it.
  bool uniform = (env_flags & 256) != 0;
  if (!uniform && !(env_flags & (16 | 512)) && !occ1)   // (whatever "gram_pace" says: the option must not change a bit of G)
    uniform = gram_choose(ctx->num_cus, (long)K, (long)n, rho_pct / 100.0).form == 2;
  if (((env_flags & 16) || uniform) && n_off >= 1) {
    GramSegs g;
    GramReduceParams rp;
    if (uniform) set_pace(218);
    // the helper plan's slots do not move through K at one rate: pacing them cost 70 %
    if (!uniform && !(env_flags & 64)) p.pace = nullptr;
    int nslabs = 0, s_off = 0;
    if (gram_plan(slots_max, tiles1, (int)(p.kx / KB), rho_pct / 100.0, g, rp.first, rp.count, &nslabs, &s_off, uniform)) {
      void* slab = nullptr;
      int rc = rt_scratch(ctx, sizeof(double) * BT * BT * 8 * (size_t)nslabs, &slab);
      if (rc != RT_OK) return rc;
      p.slab = static_cast<double*>(slab); p.nslots = nslabs;
      rp.slab_off = rp.slab_diag = p.slab; rp.G = G; rp.n = n; rp.last0 = p.last0;
      rp.nslots_off = rp.nslots_diag = nslabs; rp.tiles1 = tiles1; rp.pace = p.pace;
      if (ctx->profile) {
        if (!ctx->ev0) {
          RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev0));
          RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev1));
        }
        if (!ctx->gev0) {
          RT_HIP_CHECK(ctx, hipEventCreate(&ctx->gev0));
          RT_HIP_CHECK(ctx, hipEventCreate(&ctx->gev1));
        }
        RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        RT_HIP_CHECK(ctx, hipEventRecord(ctx->gev0, ctx->stream));
      }
      rc = kc ? launch_gram_merged<true>(ctx, p, g, 8 * slots_max) : launch_gram_merged<false>(ctx, p, g, 8 * slots_max);
      if (rc != RT_OK) return rc;
      if (ctx->profile) {
        RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        RT_HIP_CHECK(ctx, hipEventRecord(ctx->gev1, ctx->stream));
        ctx->ev_valid = true;
        ctx->gev_valid = true;
      }
      ctx->last_grid = 8 * slots_max; ctx->last_splits = 8 * s_off; ctx->last_tile = 128 * 1000 + 128;
      const long total = n * n;
      hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, rp);
      RT_HIP_CHECK(ctx, hipGetLastError());
      return RT_OK;
    }
  }

  set_pace(228);
  // ---- two launches -------------------------------------------------------------------------------------
  int S_off = n_off ? slots_max / n_off : 0;   // sub-splits per off-diagonal tile per XCD
  int S_diag = slots_max / tiles1;             // ... per diagonal tile
  // Shorter snapshot sets get fewer sub-splits, at least 48 stages each (below that the slab traffic and the two
  // launches dominate), as long as an XCD still has 16 workgroups of the bigger launch to run: 1e5 x 256 then takes
  // 197 us instead of the generic symmetric GEMM's 255, 2e5 x 128 93 instead of 286, 5e4 x 512 296 instead of 374;
  // 3e4 x 384 (12 workgroups per XCD) is left to the generic kernel, which is faster there (188 vs 212 us).
  {
    const int cap = (int)(p.kx / KB / 48);
    if (S_off > cap) S_off = cap;
    if (S_diag > cap) S_diag = cap;
    const int busiest = n_off ? n_off * S_off : tiles1 * S_diag;
    if (cap < 1 || busiest < 16) return RT_ERR_UNSUPPORTED;
  }
  const int nslots_off = S_off * n_off, nslots_diag = S_diag * tiles1;
  const size_t tile_bytes = sizeof(double) * BT * BT;
  void* slab = nullptr;
  int rc = rt_scratch(ctx, tile_bytes * 8 * (size_t)(nslots_off + nslots_diag), &slab);
  if (rc != RT_OK) return rc;
  double* slab_off = static_cast<double*>(slab);
  double* slab_diag = slab_off + (size_t)8 * nslots_off * BT * BT;

  GramReduceParams rp;
  rp.slab_off = slab_off; rp.slab_diag = slab_diag; rp.G = G; rp.n = n; rp.last0 = p.last0;
  rp.nslots_off = nslots_off; rp.nslots_diag = nslots_diag; rp.tiles1 = tiles1; rp.pace = ctx->gram_pace;

  if (ctx->profile) {
    if (!ctx->ev0) {
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev0));
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->ev1));
    }
    if (!ctx->gev0) {
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->gev0));
      RT_HIP_CHECK(ctx, hipEventCreate(&ctx->gev1));
    }
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->gev0, ctx->stream));
  }
  // launch 1: off-diagonal tiles
  if (n_off) {
    int slot = 0, t = 0;
    for (int a = 0; a < tiles1; ++a)
      for (int b = a; b < tiles1; ++b, ++t) {
        if (a == b) continue;
        rp.first[t] = (unsigned char)slot; rp.count[t] = (unsigned char)S_off;
        for (int q = 0; q < S_off; ++q, ++slot) {
          p.slot_tm[slot] = (unsigned char)a; p.slot_tn[slot] = (unsigned char)b;
          p.slot_q[slot] = (unsigned char)q; p.slot_S[slot] = (unsigned char)S_off;
        }
      }
    p.slab = slab_off; p.nslots = nslots_off;
    rc = kc ? launch_gram<true, false>(ctx, p, 8 * nslots_off) : launch_gram<false, false>(ctx, p, 8 * nslots_off);
    if (rc != RT_OK) return rc;
  }
  // launch 2: diagonal tiles (upper MFMA tiles only)
  {
    int slot = 0, t = 0;
    for (int a = 0; a < tiles1; ++a)
      for (int b = a; b < tiles1; ++b, ++t) {
        if (a != b) continue;
        rp.first[t] = (unsigned char)slot; rp.count[t] = (unsigned char)S_diag;
        for (int q = 0; q < S_diag; ++q, ++slot) {
          p.slot_tm[slot] = (unsigned char)a; p.slot_tn[slot] = (unsigned char)a;
          p.slot_q[slot] = (unsigned char)q; p.slot_S[slot] = (unsigned char)S_diag;
        }
      }
    p.slab = slab_diag; p.nslots = nslots_diag;
    p.pace = (p.pace && (env_flags & 64)) ? ctx->gram_pace + 128 : nullptr;   // every panel has one reader here
    rc = kc ? launch_gram<true, true>(ctx, p, 8 * nslots_diag) : launch_gram<false, true>(ctx, p, 8 * nslots_diag);
    if (rc != RT_OK) return rc;
  }
  if (ctx->profile) {
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->gev1, ctx->stream));
    ctx->ev_valid = true;
    ctx->gev_valid = true;
  }
  ctx->last_grid = 8 * (nslots_off + nslots_diag); ctx->last_splits = 8 * S_off; ctx->last_tile = 128 * 1000 + 128;
  const long total = n * n;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, rp);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

// Diagnostic / test hook: the form rt_gram would take (gram_choose) - out = {form, a, b, S_off, S_diag}.  No GPU needed.
extern "C" int rt_gram_plan_info(int num_cus, int64_t n_rows, int64_t n_cols, int* out) {
  if (!out || num_cus < 8 || n_rows < 1 || n_cols < 1) return RT_ERR_ARG;
  const GramChoice c = gram_choose(num_cus, (long)n_rows, (long)n_cols, 0.60);
  out[0] = c.form; out[1] = c.a; out[2] = c.b; out[3] = c.s_off; out[4] = c.s_diag;
  return RT_OK;
}
