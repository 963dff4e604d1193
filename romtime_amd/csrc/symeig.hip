// Symmetric eigensolver for the small n x n Gram matrix of the POD (3 <= n <= 1024), on the device:
// all eigenvalues + the k leading eigenvectors, so that `orth` never leaves the GPU for the step
// that LAPACK dsyevd (19 ms at n = 512 on the host, 13 ms in rocSOLVER) would otherwise dominate.
//
//   1. symeig_tridiag_kernel   Householder tridiagonalisation Q^T G Q = T (LAPACK dsytd2 recurrences).
//      32 workgroups x 1024 threads (128 for n > 512) keep G in LDS, rows dealt cyclically (row i ->
//      workgroup i % 32), so every step's mat-vec and rank-2 update run out of LDS.  ONE inter-workgroup
//      hand-off per column (the slices of p = tau A v plus the next row, pre-update) in the write-through
//      form of the guide's inter-workgroup protocol: payload stored and loaded with agent-scope relaxed
//      atomics (global_store/load ... sc1), every storing wave drains vmcnt, workgroup barrier, ONE lane
//      bumps a counter; ONE lane polls with s_sleep and a wall-clock bound, the others wait at a
//      workgroup barrier.  No fences, no dependence on placement; a timeout raises an error word and
//      every workgroup leaves.
//   2. symeig_bisect_kernel    eigenvalues of T by Sturm-count multisection: two waves per eigenvalue,
//      257 sections per pass, 7 passes (257^7 > 2^53).
//   3. symeig_wy_kernel + symeig_vectors_kernel   one wave per wanted eigenvector: inverse iteration on
//      T - lambda I (pivoted tridiagonal LU, as dstein) in LDS, then the reflectors applied in reverse,
//      four per reduction round.
// The caller (pod.py) adds a k x k Rayleigh-Ritz step on G when kept eigenvalues are clustered, which also
// cross-checks the Ritz values against the multisection eigenvalues; a mismatch raises (nothing silent).
#include <cstdlib>

#include "common.h"
#include "wave_ops.h"

namespace {

constexpr int TW_SMALL = 32;   // workgroups of the tridiagonalisation, n <= 512 (rows dealt cyclically)
constexpr int SLOT0 = 160;     // first arrival slot in the flag words (128-byte aligned)
constexpr int TW_LARGE = 128;  // 512 < n <= 1024: 8 rows of 1024 per workgroup
constexpr int LPR = 64;     // lanes per matrix row in the mat-vec / rank-2 update (TT / LPR rows per pass)
constexpr int TT = 1024;    // threads per workgroup
static_assert(LPR == 64, "row reductions use the whole-wave DPP sum");
constexpr int NMAX = 1024;     // largest supported n; the kernels are instantiated for 128, 256, 512 and 1024
// inverse iterations per eigenvector: the shifts are eigenvalues to full precision, so the first solve already
// amplifies the wanted direction by ~1/eps and the second removes what is left of the start vector (dstein also
// stops after two or three); the caller's Rayleigh-Ritz step cross-checks the result
constexpr int VEC_ITERS = 2;

struct TriParams {
  const double* G;   // n x n row-major (read only)
  double* V;         // n x n: row k = Householder vector of step k (entries j > k), sc1 traffic
  double* P;         // 4 x n: slices of p (double buffered) and the look-ahead row (double buffered)
  double* tau;       // n
  double* d;         // n
  double* e;         // n
  int* flags;        // [3] = error, [4] = worker tickets of the one-XCD form, [8 .. 8 + tw) = XCC id of each workgroup, [SLOT0 .. SLOT0 + tw) = arrival slots
  int n;
  int tw;            // cooperating workgroups
  int local;         // 1: ONE workgroup holds the whole matrix (n <= 128): p and the next row never leave its LDS
  int spread;        // 1: launched as 8 tw blocks, only the blocks with blockIdx % 8 == xcd work (they share an XCD)
  int xcd;
  long* counters;    // rt_ctx::dev_counters: which hand-off form ran, and time-outs (rt_ctx_get_counter)
};

__device__ __forceinline__ void st_wt(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One-XCD form of the hand-off (every cooperating workgroup was SEEN to run on the same XCD, see the kernel): a
// plain store leaves the line in that XCD's L2, where the sc1 loads of the readers (L1-bypassing, L2-served) find
// it - a leg costs an L2 round trip instead of a memory-side one.
__device__ __forceinline__ void st_xcd(double* p, double v, bool one_xcd) {
  if (one_xcd)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // a plain global_store
  else
    st_wt(p, v);
}
__device__ __forceinline__ void st_slot(int* p, int v, bool one_xcd) {
  if (one_xcd)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_wt(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Arrival at a hand-off = storing the hop number into the workgroup's own slot; one wave waits until every slot
// holds >= want (the TW slots are one or two 128-byte lines: one load per poll, and no read-modify-write that 32
// arrivals would queue up behind - a shared counter cost 0.6 us per hop more).  Called by a whole wave; false if
// the error word is set or 0.5 s pass.
__device__ __forceinline__ bool poll_slots(int* slots, int TW, int want, int* err) {
  const int lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  for (;;) {
    int f = want;
    for (int w = lane; w < TW; w += 64) {
      const int g = __hip_atomic_load(&slots[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      f = g < f ? g : f;
    }
    if (__builtin_amdgcn_ballot_w64(f < want) == 0) return true;
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {  // 0.5 s at 100 MHz
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}

__device__ __forceinline__ double block_sum(double x, double* s_red) {
  x = rtw::wave_sum(x);
  const int wid = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[wid] = x;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < TT / 64; ++w) t += s_red[w];  // same order in every thread
  return t;
}

// One inter-workgroup hand-off per column: together with its slice of p = tau A v, the owner of row
// k+1 publishes that row as it stands BEFORE the rank-2 update of step k; after the hand-off every
// workgroup applies the update to its copy and derives the next reflector redundantly (and, the
// arithmetic being identical, bit-identically).  All O(n) vector work of a step (reflector, p.v, w, the
// look-ahead row) is done by wave 0 alone in registers with DPP reductions - NM / 64 elements per lane -
// so a step costs two or three workgroup barriers; the other 15 waves only do the O(n^2/P) mat-vec and update.
template <int NM>
__global__ __launch_bounds__(TT) void symeig_tridiag_kernel(const TriParams p) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  // One-XCD form: of the 16 tw blocks launched, the first tw that find themselves on XCD p.xcd become the workers
  // (a ticket each; the rest, and everything that landed elsewhere, leave at once).  Where a block lands is observed,
  // not derived from blockIdx: with other kernels in flight - eigensolves of other contexts on other XCDs,
  // romtime_amd.pipeline.PodLanes - block i is not on XCD i % 8.
  __shared__ int s_ticket;
  if (p.spread) {
    if (threadIdx.x == 0) {
      const int xcc = (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | (0 << 6) | 20) & 15u);
      s_ticket = (xcc == p.xcd) ? __hip_atomic_fetch_add(&p.flags[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.tw;
    }
    __syncthreads();
    if (s_ticket >= p.tw) return;
  }
  const int n = p.n, TW = p.tw, tid = threadIdx.x, wg = p.spread ? s_ticket : blockIdx.x, lane = tid & 63,
            wid = tid >> 6;
  const int RB = (n + TW - 1) / TW;
  const bool local = p.local != 0;   // wave-uniform: tw == 1, every row is this workgroup's
  // local form: eight lanes share a row (see the mat-vec), so eight rows are read at once - two words of padding per row
  // spread them over the banks
  const int lda = local ? n + 2 : n;
  double* A = sm;                 // RB x lda, local row li <-> global row li*TW + wg
  double* svb = sm + (size_t)RB * lda;  // v, double buffered by step parity (2 x n)
  double* sw = svb + 2 * n;
  double* sl = sw + n;            // current row k (entries j >= k), maintained by wave 0
  double* sP = sl + n;            // local form only: p = tau A v (the host sizes the LDS for it)
  __shared__ int s_abort;
  if (threadIdx.x == 0) s_abort = 0;
  constexpr int PER = NM / 64;

  for (int q = tid; q < RB * n; q += TT) {
    const int li = q / n, j = q % n, gi = li * TW + wg;
    A[(size_t)li * lda + j] = (gi < n) ? p.G[(size_t)gi * n + j] : 0.0;
  }
  __syncthreads();
  const int ty = tid / LPR, tx = tid % LPR;  // (TT / LPR) rows x LPR lanes; with LPR = 64: ty == wave id
  int hop = 0;                                // hand-offs so far

  // step "-1": row 0 to everybody
  // P = [p even | p odd | row even | row odd].  Row 0 travels in the ODD row buffer: step 0 stores into the even
  // one, and a workgroup may get there before a slow one has read row 0.
  double* L0 = p.P + (size_t)3 * n;
  __shared__ int s_one;
  if (local) {
    // Round 3: a Gram matrix of up to 128 columns fits ONE workgroup's LDS.  A column then needs no hand-off at all -
    // no payload stores to drain, no arrival slot, no poll, no payload loads: each of those is a memory round trip of
    // 0.5-0.7 us that the cooperative form pays even when the "team" is a single workgroup (3.3 us per column at n = 64
    // whatever the team size).  p goes through LDS (sP), the next row is read where it lies.
    for (int j = tid; j < n; j += TT) sl[j] = A[j];
    if (tid == 0) {
      s_one = 0;
      atomicAdd(reinterpret_cast<unsigned long long*>(&p.counters[RT_CNT_EIG_ONE_XCD]), 1ull);
    }
    __syncthreads();
  } else {
  if (wg == 0)
    for (int j = tid; j < n; j += TT) st_wt(&L0[j], A[j]);
  if (tid == 0)  // where this workgroup runs: HW_REG_XCC_ID[3:0]
    __hip_atomic_store(&p.flags[8 + wg], (int)__builtin_amdgcn_s_getreg((4 - 1) << 11 | (0 << 6) | 20) + 1,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ++hop;
  if (wid == 0) {
    if (lane == 0) st_slot(&p.flags[SLOT0 + wg], hop, false);
    if (!poll_slots(&p.flags[SLOT0], TW, hop, &p.flags[3]) && lane == 0) s_abort = 1;
  }
  __syncthreads();
  if (s_abort) return;
  for (int j = tid; j < n; j += TT) sl[j] = ld_wt(&L0[j]);
  // every workgroup reads the same TW ids and takes the same decision; the placement is observed, not assumed
  if (tid == 0) {
    const int mine = __hip_atomic_load(&p.flags[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int same = (p.spread != 0) && (mine > 0);
    for (int w = 1; w < TW; ++w)
      same &= (__hip_atomic_load(&p.flags[8 + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine);
    s_one = same;
    if (wg == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&p.counters[same ? RT_CNT_EIG_ONE_XCD : RT_CNT_EIG_GENERAL_FORM]), 1ull);
  }
  __syncthreads();
  }  // !local
  const bool one_xcd = s_one != 0;

  __shared__ double s_tauv[2];  // tau of step k in s_tauv[k & 1]
  // n > 512 (128 workgroups, few rows each): the next step's reflector is formed by wave 0 in the same section as w
  // and the updated row, straight from registers - two workgroup barriers per column, 10 % faster.  n <= 512: the
  // reflector is formed after the rank-2 update has been released (three barriers); there the merged form measured
  // 5 % slower.
  constexpr bool MERGED = NM > 512;
  // Reflector (dlarfg) of step kk from the registers of wave 0: x[q] = element j0 + lane + 64 q of row kk, the
  // reflector starts `off` elements in (alpha = that element, v there = 1).  Identical data and arithmetic in every
  // workgroup.  Writes v to the step's LDS buffer, tau to s_tauv, and (owner only) V, tau, d, e to memory.
  auto reflector = [&](const double (&x)[PER], int kk, int j0, int off, double diag) {
    double part = 0.0;
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (q > 0 || lane > off) part = fma(x[q], x[q], part);  // x is 0 beyond the row's end
    part = rtw::wave_sum(part);
    const double alpha = rtw::read_lane(x[0], off);
    double tau, beta, scale;
    if (part == 0.0) {
      tau = 0.0; beta = alpha; scale = 0.0;
    } else {
      beta = -copysign(sqrt(alpha * alpha + part), alpha);
      tau = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    const bool owner = (wg == kk % TW);
    double* svk = svb + (size_t)(kk & 1) * n;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int j = j0 + lane + 64 * q;
      const double vq = (q == 0 && lane == off) ? 1.0 : x[q] * scale;
      if (j < n && (q > 0 || lane >= off)) {
        svk[j] = vq;
        if (owner) p.V[(size_t)kk * n + j] = vq;
      }
    }
    if (lane == 0) {
      s_tauv[kk & 1] = tau;
      if (owner) {
        p.tau[kk] = tau;
        p.d[kk] = diag;
        p.e[kk] = beta;
      }
    }
  };

  // A step is: mat-vec (all waves) -> hand-off -> wave 0 alone: w, the updated next row AND from it the next
  // step's reflector, all in registers -> rank-2 update (all waves)  [MERGED; otherwise the reflector follows the update].
  if (wid == 0 && n > 2) {
    double x[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int j = 1 + lane + 64 * q;
      x[q] = (j < n) ? sl[j] : 0.0;
    }
    reflector(x, 0, 1, 0, sl[0]);
  }
  __syncthreads();
  for (int k = 0; k + 2 < n; ++k) {
    const double* sv = svb + (size_t)(k & 1) * n;
    if constexpr (!MERGED) {
      if (k > 0) {
        if (wid == 0) {  // reflector of row k from the row as wave 0 left it in LDS
          double x[PER];
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            const int j = k + 1 + lane + 64 * q;
            x[q] = (j < n) ? sl[j] : 0.0;
          }
          reflector(x, k, k + 1, 0, sl[k]);
        }
        __syncthreads();  // B1
      }
    }
    const double tau = s_tauv[k & 1];
    // ---- p = tau * A v on the local rows i > k; the wave that holds row k+1 adds that row (pre-update) ----
    double* Pk = p.P + (size_t)(k & 1) * n;
    double* Lk = p.P + (size_t)(2 + (k & 1)) * n;
    if (local) {
      // 128 rows x 8 lanes: every row of the matrix is reduced at once (a wave per row took n / 16 rounds of a 64-lane
      // DPP sum one after the other: the mat-vec, not the hand-off, would be the column's time here)
      const int gi = tid >> 3, sub = tid & 7;
      double acc = 0.0;
      if (tau != 0.0 && gi > k && gi < n) {
        const double* row = A + (size_t)gi * lda;
        for (int j = k + 1 + sub; j < n; j += 8) acc = fma(row[j], sv[j], acc);
      }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      acc += __shfl_xor(acc, 4);
      if (sub == 0 && tau != 0.0 && gi > k && gi < n) sP[gi] = tau * acc;
    } else
    for (int li = ty; li < RB; li += TT / LPR) {
      const int gi = li * TW + wg;
      if (gi > k && gi < n) {
        const double* row = A + (size_t)li * n;
        if (tau != 0.0) {
          double acc = 0.0;
          for (int j = k + 1 + tx; j < n; j += LPR) acc = fma(row[j], sv[j], acc);
          acc = rtw::wave_sum(acc);
          if (tx == 0) {
            if (local) sP[gi] = tau * acc;
            else st_xcd(&Pk[gi], tau * acc, one_xcd);
          }
        }
        if (gi == k + 1 && !local)
          for (int j = k + 1 + tx; j < n; j += LPR) st_xcd(&Lk[j], row[j], one_xcd);
      }
    }
    if (!local) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // B2
    // the hop numbers only grow: nobody can arrive for hop h+1 before all arrived for hop h, and the payload
    // buffers alternate, so a fast workgroup never overwrites what a slow one still reads.  Wave 0 arrives, waits
    // and goes straight on to the payload; the other waves meet it again at B5.
    ++hop;
    if (wid == 0) {
      if (lane == 0 && !local) st_slot(&p.flags[SLOT0 + wg], hop, one_xcd);
      double v[PER];  // this step's reflector, element j = k + 1 + lane + 64 q (LDS read in flight during the poll)
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int j = k + 1 + lane + 64 * q;
        v[q] = (j < n) ? sv[j] : 0.0;
      }
      if (!local && !poll_slots(&p.flags[SLOT0], TW, hop, &p.flags[3])) {
        if (lane == 0) s_abort = 1;
      } else {
        // wave 0 takes p and the next row straight into registers
        double pv[PER], x[PER];
        const double* nextrow = A + (size_t)(k + 1) * lda;   // local form: row k + 1 as it stands, pre-update
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int j = k + 1 + lane + 64 * q;
          if (local) {
            x[q] = (j < n) ? nextrow[j] : 0.0;
            pv[q] = (j < n && tau != 0.0) ? sP[j] : 0.0;
          } else {
            x[q] = (j < n) ? ld_wt(&Lk[j]) : 0.0;
            pv[q] = (j < n && tau != 0.0) ? ld_wt(&Pk[j]) : 0.0;
          }
        }
        if (tau != 0.0) {
          double dot = 0.0;
#pragma unroll
          for (int q = 0; q < PER; ++q) dot = fma(pv[q], v[q], dot);
          dot = rtw::wave_sum(dot);
          const double alpha2 = -0.5 * tau * dot;
#pragma unroll
          for (int q = 0; q < PER; ++q) pv[q] = fma(alpha2, v[q], pv[q]);  // pv now holds w
          const double w1 = rtw::first_lane(pv[0]);                        // w_{k+1}; v_{k+1} = 1
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            const int j = k + 1 + lane + 64 * q;
            x[q] = (j < n) ? x[q] - (pv[q] + w1 * v[q]) : 0.0;  // row k+1 after the rank-2 update
            if (j < n) sw[j] = pv[q];
          }
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int j = k + 1 + lane + 64 * q;
          if (j < n) sl[j] = x[q];  // read again only for the trailing 2 x 2
        }
        if constexpr (MERGED)
          if (k + 3 < n) reflector(x, k + 1, k + 1, 1, rtw::first_lane(x[0]));
      }
    }
    __syncthreads();  // B5: sw, the next step's v and tau
    if (s_abort) return;
    if (tau == 0.0) continue;
    // ---- A <- A - v w^T - w v^T on the local rows i > k (each wave touches only the rows it also reads) ----
    if (local) {
      const int gi = tid >> 3, sub = tid & 7;
      if (gi > k && gi < n) {
        double* row = A + (size_t)gi * lda;
        const double vi = sv[gi], wi = sw[gi];
        for (int j = k + 1 + sub; j < n; j += 8) row[j] -= vi * sw[j] + wi * sv[j];
      }
    } else
    for (int li = ty; li < RB; li += TT / LPR) {
      const int gi = li * TW + wg;
      if (gi > k && gi < n) {
        double* row = A + (size_t)li * n;
        const double vi = sv[gi], wi = sw[gi];
        for (int j = k + 1 + tx; j < n; j += LPR) row[j] -= vi * sw[j] + wi * sv[j];
      }
    }
  }
  __syncthreads();
  // trailing 2 x 2: sl holds row n-2 (entries n-2, n-1); row n-1 is with its owner
  if (wg == (n - 2) % TW && tid == 0) {
    p.d[n - 2] = sl[n - 2];
    p.e[n - 2] = sl[n - 1];
  }
  if (wg == (n - 1) % TW && tid == 0) {
    p.d[n - 1] = A[(size_t)((n - 1) / TW) * lda + (n - 1)];
    p.e[n - 1] = 0.0;
  }
}

// 1/q by v_rcp_f64 and one Newton step (<= 1 ulp): the IEEE divide sequence is ~4x as long and the
// count only has to be that of a matrix within eps of T
__device__ __forceinline__ double fast_rcp(double q) {
  const double r = __builtin_amdgcn_rcp(q);
  return fma(fma(-q, r, 1.0), r, r);
}

__device__ __forceinline__ int sturm_count(const double* d, const double* e2, int n, double x, double pivmin) {
  double q = d[0] - x;
  if (fabs(q) < pivmin) q = -pivmin;
  int c = (q < 0.0);
  for (int i = 1; i < n; ++i) {
    q = fma(-e2[i - 1], fast_rcp(q), d[i] - x);
    if (fabs(q) < pivmin) q = -pivmin;
    c += (q < 0.0);
  }
  return c;
}

__global__ void symeig_init_kernel(int* flags, double* tau, int n) {
  for (int i = threadIdx.x; i < SLOT0 + TW_LARGE; i += blockDim.x) flags[i] = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) tau[i] = 0.0;
}

// lam[j] = j-th LARGEST eigenvalue of T by Sturm-count multisection, 257 sections per pass, 7 passes
// (257^7 > 2^53).  The loop is instruction-issue bound (one wave per SIMD, ~9 FP64 instructions per shift and row),
// so an eigenvalue is shared by TWO waves with 2 shifts per lane, and a workgroup is four waves = two
// eigenvalues: its waves occupy the four SIMDs of one CU, n/2 workgroups put one wave on every SIMD of the chip.
constexpr int BIS_NS = 2;  // shifts per lane
template <int NM>
__global__ __launch_bounds__(256) void symeig_bisect_kernel(const double* __restrict__ d, const double* __restrict__ e,
                                                            int n, int first, int count, double* __restrict__ lam,
                                                            const int* flags, int* status, long* counters) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (status) *status = flags[3];
    if (flags[3]) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[RT_CNT_EIG_TIMEOUT]), 1ull);
  }
  __shared__ double sd[NM], se2[NM];
  __shared__ int s_first[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = wid & 1;
  double glo = 1e300, ghi = -1e300, emax = 0.0;
  for (int i = lane; i < n; i += 64) {  // every wave scans all of T (same values): its bounds are complete
    const double di = d[i], ei = (i + 1 < n) ? e[i] : 0.0, el = (i > 0) ? fabs(e[i - 1]) : 0.0;
    sd[i] = di;
    se2[i] = ei * ei;
    glo = fmin(glo, di - el - fabs(ei));
    ghi = fmax(ghi, di + el + fabs(ei));
    emax = fmax(emax, ei * ei);
  }
  glo = rtw::wave_min(glo);
  ghi = rtw::wave_max(ghi);
  emax = rtw::wave_max(emax);
  __syncthreads();
  const double tn = fmax(fabs(glo), fabs(ghi));
  double lo = glo - 2.2e-16 * tn * n - 1e-300, hi = ghi + 2.2e-16 * tn * n + 1e-300;
  const double pivmin = 2.2250738585072014e-308 * fmax(1.0, emax);
  // descending index, inside [first, first + count) (an odd count repeats the last one)
  const int j = first + min(2 * (int)blockIdx.x + (wid >> 1), count - 1);
  const int want = n - j;        // ascending rank (1-based): smallest x with count(x) >= want
  constexpr int NSH = 128 * BIS_NS;  // shifts per pass
  constexpr int NSEC = NSH + 1;
  for (int pass = 0; pass < 7; ++pass) {
    const double h = (hi - lo) / NSEC;
    const int s0 = BIS_NS * (64 * half + lane);
    const double x0 = lo + h * (s0 + 1), x1 = x0 + h;
    double q0 = sd[0] - x0, q1 = sd[0] - x1;
    if (fabs(q0) < pivmin) q0 = -pivmin;
    if (fabs(q1) < pivmin) q1 = -pivmin;
    int c0 = (q0 < 0.0), c1 = (q1 < 0.0);
    for (int i = 1; i < n; ++i) {
      const double di = sd[i], ei = se2[i - 1];
      q0 = fma(-ei, fast_rcp(q0), di - x0);
      q1 = fma(-ei, fast_rcp(q1), di - x1);
      if (fabs(q0) < pivmin) q0 = -pivmin;
      if (fabs(q1) < pivmin) q1 = -pivmin;
      c0 += (q0 < 0.0); c1 += (q1 < 0.0);
    }
    int first = NSH;  // index of the first shift whose count reaches `want` (none: NSH)
    if (c1 >= want) first = s0 + 1;
    if (c0 >= want) first = s0;
    first = rtw::wave_min_i32(first);
    if (lane == 0) s_first[pass & 1][wid] = first;
    __syncthreads();  // this parity's slots are rewritten two barriers later
    first = min(s_first[pass & 1][wid & 2], s_first[pass & 1][(wid & 2) + 1]);
    if (first == NSH) {
      lo = lo + h * NSH;
    } else {
      hi = lo + h * (first + 1);
      lo = lo + h * first;
    }
  }
  if (lane == 0 && half == 0) lam[j] = 0.5 * (lo + hi);
}

struct VecParams {
  const double* d;
  const double* e;
  const double* lam;   // descending
  const double* V;     // reflectors
  const double* tau;
  const double* C;     // 6 cross products v_u . v_w per block of 4 reflectors (symeig_wy_kernel)
  double* W;           // n x k row-major output: column t = eigenvector of lam[t]
  int n, k;
};

constexpr int RBK = 4;  // reflectors applied together in the back-transformation

// Block b holds the reflectors k0 - u, u = 0..3, k0 = n - 3 - 4 b, applied in that order.  Their mutual inner
// products do not depend on the vector being transformed, so they are computed once here (one wave per block)
// and shared by all eigenvectors: C[6 b + {0..5}] = v1.v0, v2.v0, v2.v1, v3.v0, v3.v1, v3.v2.
template <int NM>
__global__ __launch_bounds__(64) void symeig_wy_kernel(const double* __restrict__ V, int n, double* __restrict__ C) {
  constexpr int PER = NM / 64;
  const int b = blockIdx.x, lane = threadIdx.x, k0 = n - 3 - RBK * b;
  double v[RBK][PER];
#pragma unroll
  for (int u = 0; u < RBK; ++u)
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int k = k0 - u, j = lane + 64 * q;
      v[u][q] = (k >= 0 && j > k && j < n) ? V[(size_t)k * n + j] : 0.0;
    }
  int slot = 0;
#pragma unroll
  for (int u = 1; u < RBK; ++u)
#pragma unroll
    for (int w = 0; w < u; ++w) {
      double dot = 0.0;
#pragma unroll
      for (int q = 0; q < PER; ++q) dot = fma(v[u][q], v[w][q], dot);
      dot = rtw::wave_sum(dot);
      if (lane == 0) C[6 * b + slot] = dot;
      ++slot;
    }
}

// one wave per eigenvector: inverse iteration on T - lambda I, then back-transformation
template <int NM>
__global__ __launch_bounds__(64) void symeig_vectors_kernel(const VecParams p) {
  // read-only inputs of the serial sweeps (dd, ee, and y / z alternately) and write-only outputs (ra, ub, ud, lm,
  // piv) live in DIFFERENT arrays, so the loads of later steps can be issued ahead of the dependent arithmetic:
  // the chain per step is then a reciprocal and three FMAs instead of an LDS round trip per access
  extern __shared__ __attribute__((aligned(16))) double vsm[];  // 8 arrays of NM (+1) doubles and NM pivot flags
  double *dd = vsm, *ee = dd + NM, *ra = ee + NM + 8, *ub = ra + NM, *ud = ub + NM, *lm = ud + NM, *z = lm + NM,
         *y = z + NM;
  unsigned char* piv = reinterpret_cast<unsigned char*>(y + NM);
  const int n = p.n, t = blockIdx.x, lane = threadIdx.x;
  const double lambda = p.lam[t];
  double tnorm = 0.0;
  for (int i = lane; i < n; i += 64) {
    dd[i] = p.d[i] - lambda;
    ee[i] = (i + 1 < n) ? p.e[i] : 0.0;   // sub- and superdiagonal (symmetric)
    tnorm = fmax(tnorm, fabs(p.d[i]) + 2.0 * ((i + 1 < n) ? fabs(p.e[i]) : 0.0));
    // deterministic start vector in (-1, 1), different for every eigenvector
    unsigned s = 1664525u * (unsigned)(i + 1 + 7919 * (t + 1)) + 1013904223u;
    s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    z[i] = (double)(s & 0xffffff) / 8388608.0 - 1.0;
  }
  if (lane == 0) ee[n] = 0.0;
  tnorm = rtw::wave_max(tnorm);
  const double tiny = 2.2e-16 * fmax(tnorm, 1e-300);
  __syncthreads();
  if (lane == 0) {
    // P L U = T - lambda I with partial pivoting (dlagtf recurrences); the running diagonal and superdiagonal
    // entries stay in registers, ra[] receives 1/pivot
    double ak = dd[0], bk = ee[0];
#pragma unroll 4
    for (int k = 0; k + 1 < n; ++k) {
      const double ck = ee[k], an = dd[k + 1], t2 = ee[k + 1];
      const bool swap = fabs(ck) > fabs(ak);
      double pk = swap ? ck : ak;
      if (!swap && fabs(pk) < tiny) pk = copysign(tiny, pk == 0.0 ? 1.0 : pk);
      const double rp = fast_rcp(pk);
      const double mult = (swap ? ak : ck) * rp;
      ra[k] = rp;
      lm[k] = mult;
      piv[k] = swap ? 1 : 0;
      ub[k] = swap ? an : bk;
      ud[k] = swap ? t2 : 0.0;
      const double nd = swap ? fma(-mult, an, bk) : fma(-mult, bk, an);
      bk = swap ? -mult * t2 : t2;
      ak = nd;
    }
    ra[n - 1] = fast_rcp((fabs(ak) < tiny) ? copysign(tiny, ak == 0.0 ? 1.0 : ak) : ak);
    ub[n - 1] = 0.0;
    ud[n - 1] = 0.0;
  }
  __syncthreads();
  for (int it = 0; it < VEC_ITERS; ++it) {
    double zmax = 0.0;
    if (lane == 0) {
      // forward: y = L^-1 P z (the running entry stays in a register)
      double cur = z[0];
#pragma unroll 4
      for (int k = 0; k + 1 < n; ++k) {
        const double nxt = z[k + 1], ck = lm[k];
        const bool sw = piv[k] != 0;
        y[k] = sw ? nxt : cur;
        cur = sw ? fma(-ck, nxt, cur) : fma(-ck, cur, nxt);
      }
      y[n - 1] = cur;
      // backward: U x = y with the reciprocal pivots
      double z1 = 0.0, z2 = 0.0;
#pragma unroll 4
      for (int k = n - 1; k >= 0; --k) {
        const double zk = fma(-ud[k], z2, fma(-ub[k], z1, y[k])) * ra[k];
        z[k] = zk;
        zmax = fmax(zmax, fabs(zk));
        z2 = z1;
        z1 = zk;
      }
    }
    zmax = rtw::first_lane(zmax);
    __syncthreads();
    const double sc = 1.0 / fmax(zmax, 1e-300);
    for (int i = lane; i < n; i += 64) z[i] *= sc;   // all lanes
    __syncthreads();
  }
  // unit 2-norm
  double part = 0.0;
  for (int i = lane; i < n; i += 64) part += z[i] * z[i];
  part = rtw::wave_sum(part);
  const double inv = 1.0 / sqrt(part);
  for (int i = lane; i < n; i += 64) z[i] *= inv;
  __syncthreads();
  // x = Q z = H_0 H_1 ... H_{n-3} z : apply the reflectors from the last to the first.  A reflector row comes
  // from L2 / HBM (~1-2 us away) while applying one takes ~0.1 us, so the rows are fetched RB at a time, one
  // whole block ahead of the block being applied (lane owns j = lane + 64 q).
  constexpr int PER = NM / 64;
  double zr[PER], vc[RBK][PER], vn[RBK][PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int j = lane + 64 * q;
    zr[q] = (j < n) ? z[j] : 0.0;
  }
  auto fetch = [&](int k, double (&dst)[PER]) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int j = lane + 64 * q;
      dst[q] = (k >= 0 && j > k && j < n) ? p.V[(size_t)k * n + j] : 0.0;
    }
  };
#pragma unroll
  for (int u = 0; u < RBK; ++u) fetch(n - 3 - u, vc[u]);
  int blk = 0;
  for (int k0 = n - 3; k0 >= 0; k0 -= RBK, ++blk) {
#pragma unroll
    for (int u = 0; u < RBK; ++u) fetch(k0 - RBK - u, vn[u]);
    double tk[RBK], cc[6];
#pragma unroll
    for (int u = 0; u < RBK; ++u) tk[u] = (k0 - u >= 0) ? p.tau[k0 - u] : 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) cc[q] = p.C[6 * blk + q];
    // the four reflectors k0, k0-1, k0-2, k0-3 at once: a_u = v_u . z for the incoming z (four independent
    // reductions), then s_u = tau_u (a_u - sum_{w<u} (v_u . v_w) s_w) and z -= sum_u s_u v_u.  One reduction round
    // on the serial chain per block instead of one per reflector.
    double a[RBK];
#pragma unroll
    for (int u = 0; u < RBK; ++u) {
      double dot = 0.0;
#pragma unroll
      for (int q = 0; q < PER; ++q) dot = fma(vc[u][q], zr[q], dot);
      a[u] = dot;
    }
#pragma unroll
    for (int u = 0; u < RBK; ++u) a[u] = rtw::wave_sum(a[u]);
    const double s0 = tk[0] * a[0];
    const double s1 = tk[1] * (a[1] - cc[0] * s0);
    const double s2 = tk[2] * (a[2] - cc[1] * s0 - cc[2] * s1);
    const double s3 = tk[3] * (a[3] - cc[3] * s0 - cc[4] * s1 - cc[5] * s2);
#pragma unroll
    for (int q = 0; q < PER; ++q)
      zr[q] = fma(-s3, vc[3][q], fma(-s2, vc[2][q], fma(-s1, vc[1][q], fma(-s0, vc[0][q], zr[q]))));
#pragma unroll
    for (int u = 0; u < RBK; ++u)
#pragma unroll
      for (int q = 0; q < PER; ++q) vc[u][q] = vn[u][q];
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int j = lane + 64 * q;
    if (j < n) z[j] = zr[q];
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) p.W[(size_t)i * p.k + t] = z[i];
}

}  // namespace

extern "C" int rt_sym_eig_values(rt_ctx* ctx, const double* G, int64_t n, double* lam, int* status) {
  return rt_sym_eig_values_part(ctx, G, n, 0, n, lam, status);
}

extern "C" int rt_sym_eig_values_part(rt_ctx* ctx, const double* G, int64_t n, int64_t first, int64_t count, double* lam,
                                      int* status) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, G && lam && n >= 3 && first >= 0 && count >= 1 && first + count <= n);
  if (n > NMAX) {
    ctx->err = "rt_sym_eig_values: n > 1024 not supported (the matrix must stay resident in the LDS of <= 128 CUs)";
    return RT_ERR_UNSUPPORTED;
  }
  const bool large = n > 512;
  static const int eig_flags = [] { const char* e = getenv("ROMTIME_EIG_FLAGS"); return e ? atoi(e) : 0; }();
  // A ctx confined to few CUs (CU-masked stream of the POD pipeline, "cu_limit") halves the team for n <= 512: 16
  // workgroups hold 32 rows each (147 KB of LDS); a column then costs more mat-vec but the same hand-off.
  const bool small_team = !large && ctx->num_cus < TW_SMALL && ctx->num_cus >= TW_SMALL / 2;
  static const int tw_env = [] { const char* e = getenv("ROMTIME_EIG_TW"); return e ? atoi(e) : 0; }();   // measurement switch
  int tw = large ? TW_LARGE : (small_team ? TW_SMALL / 2 : TW_SMALL);
  if (!large && tw_env >= 1 && tw_env <= TW_SMALL && (long)((n + tw_env - 1) / tw_env) * n * 8 <= 120 * 1024) tw = tw_env;
  // One workgroup holds the whole matrix: no hand-offs (see the kernel).  Measured (tools/probes/eig_tw_ab.py, values + 40
  // vectors): n = 16 / 33 / 64: 0.076 / 0.120 / 0.211 ms against 0.091 / 0.140 / 0.233 in the cooperative form - but
  // n = 128: 0.50 against 0.44: the hand-off is NOT what a column costs at these sizes (nor does the team size matter:
  // 32, 16, 8, 2 or 1 workgroups give the same 3.3 us per column); it is the three barriers of a 1024-thread workgroup and
  // wave 0's serial sections, and with all 128 rows in one workgroup those get longer.  ROMTIME_EIG_FLAGS & 8 extends the
  // form to n <= 128 for measurements.
  const bool local = (n <= 64 || ((eig_flags & 8) && n <= 128)) && !(eig_flags & 4);
  if (local) tw = 1;
  if (tw > ctx->num_cus) {  // the cooperating workgroups must all be resident
    ctx->err = "rt_sym_eig_values: not enough compute units for the cooperative tridiagonalisation";
    return RT_ERR_UNSUPPORTED;
  }
  // composite arena: V (n*n) | P (2n) | tau | d | e | flags
  size_t off = 0;
  auto take = [&off](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t oV = take(sizeof(double) * n * n), oP = take(sizeof(double) * 4 * n), oT = take(sizeof(double) * n),
               oD = take(sizeof(double) * n), oE = take(sizeof(double) * n), oF = take(sizeof(int) * (SLOT0 + TW_LARGE));
  void* base = nullptr;
  int rc = rt_scratch2(ctx, off, &base);
  if (rc != RT_OK) return rc;
  char* b8 = static_cast<char*>(base);
  TriParams tp;
  tp.G = G; tp.n = (int)n; tp.tw = tw;
  // n <= 512: launch 16 x 32 blocks, of which the first 32 that find themselves on XCD eig_xcd work, so the 32
  // workers can hand off through one L2 (the kernel checks HW_REG_XCC_ID and falls back to the general form).
  // ROMTIME_EIG_FLAGS=1 disables it.
  tp.spread = (!large && !local && ctx->eig_one_xcd && !(eig_flags & 1) && ctx->num_cus / 8 >= tw) ? 1 : 0;  // a CU per worker
  tp.local = local ? 1 : 0;
  tp.xcd = ctx->eig_xcd & 7;
  tp.V = reinterpret_cast<double*>(b8 + oV); tp.P = reinterpret_cast<double*>(b8 + oP);
  tp.tau = reinterpret_cast<double*>(b8 + oT); tp.d = reinterpret_cast<double*>(b8 + oD);
  tp.e = reinterpret_cast<double*>(b8 + oE); tp.flags = reinterpret_cast<int*>(b8 + oF);
  tp.counters = ctx->dev_counters;
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(symeig_init_kernel, dim3(1), dim3(256), 0, st, tp.flags, tp.tau, (int)n);

  const int RB = (int)((n + tw - 1) / tw);
  const size_t lds = sizeof(double) * ((size_t)RB * (local ? n + 2 : n) + 5 * n + 16);  // A slab | v (x2) | w | row | p (local form)
  // n <= 256 has instantiations of its own (round 3): wave 0's register work per column - reflector, w, the updated row:
  // NM / 64 elements per lane - and the eigenvector kernel's back-transformation halve against the 512 ones
  static const bool small_nm = !(eig_flags & 2);
  const bool tiny = small_nm && n <= 256, tinier = small_nm && n <= 128;
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&symeig_tridiag_kernel<128>), 150 * 1024));
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&symeig_tridiag_kernel<256>), 150 * 1024));
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&symeig_tridiag_kernel<512>), 150 * 1024));
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&symeig_tridiag_kernel<1024>), 150 * 1024));
  {
    // The tw workers spin on each other: all of them must be resident at once.  Ask the runtime how many of these
    // workgroups a CU holds (1: the slab takes most of its LDS) and refuse up front what the device cannot hold,
    // instead of finding out through the hand-off's wall-clock bound.
    int per_cu = 0;
    if (large)
      RT_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, symeig_tridiag_kernel<1024>, TT, lds));
    else if (tinier)
      RT_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, symeig_tridiag_kernel<128>, TT, lds));
    else if (tiny)
      RT_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, symeig_tridiag_kernel<256>, TT, lds));
    else
      RT_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, symeig_tridiag_kernel<512>, TT, lds));
    if ((long)per_cu * ctx->num_cus < tw) {
      ctx->err = "rt_sym_eig_values: the device cannot keep all cooperating workgroups resident";
      return RT_ERR_UNSUPPORTED;
    }
  }
  const dim3 bgrid((unsigned)((count + 1) / 2));
  if (large) {
    hipLaunchKernelGGL(symeig_tridiag_kernel<1024>, dim3(tw), dim3(TT), lds, st, tp);
    RT_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(symeig_bisect_kernel<1024>, bgrid, dim3(256), 0, st, tp.d, tp.e, (int)n, (int)first, (int)count,
                       lam, tp.flags, status, ctx->dev_counters);
  } else {
    if (tinier)
      hipLaunchKernelGGL(symeig_tridiag_kernel<128>, dim3(tp.spread ? 16 * tw : tw), dim3(TT), lds, st, tp);
    else if (tiny)
      hipLaunchKernelGGL(symeig_tridiag_kernel<256>, dim3(tp.spread ? 16 * tw : tw), dim3(TT), lds, st, tp);
    else
      hipLaunchKernelGGL(symeig_tridiag_kernel<512>, dim3(tp.spread ? 16 * tw : tw), dim3(TT), lds, st, tp);
    RT_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(symeig_bisect_kernel<512>, bgrid, dim3(256), 0, st, tp.d, tp.e, (int)n, (int)first, (int)count,
                       lam, tp.flags, status, ctx->dev_counters);
  }
  RT_HIP_CHECK(ctx, hipGetLastError());
  ctx->eig.d = tp.d; ctx->eig.e = tp.e; ctx->eig.V = tp.V; ctx->eig.tau = tp.tau; ctx->eig.n = n; ctx->eig.base = base;
  ctx->eig.gen = ctx->scratch2_gen;
  return RT_OK;
}

extern "C" int rt_sym_eig_vectors(rt_ctx* ctx, int64_t n, int64_t k, const double* lam, double* W) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, lam && W && k >= 1 && k <= n);
  // the reflectors live in the composite arena: any operator that has carved it since (greedy, projections, the
  // sweeps - every rt_scratch2 call bumps the generation) has overwritten them
  if (ctx->eig.n != n || ctx->eig.base == nullptr || ctx->eig.base != ctx->scratch2 || ctx->eig.gen != ctx->scratch2_gen) {
    ctx->err = "rt_sym_eig_vectors: must follow rt_sym_eig_values on the same ctx and matrix with no other composite "
               "operator (greedy, projection, sweep) in between";
    return RT_ERR_ARG;
  }
  VecParams vp;
  vp.d = ctx->eig.d; vp.e = ctx->eig.e; vp.lam = lam; vp.V = ctx->eig.V; vp.tau = ctx->eig.tau;
  vp.W = W; vp.n = (int)n; vp.k = (int)k;
  const int nblk = (int)((n - 2 + RBK - 1) / RBK);
  void* cbuf = nullptr;
  int rc = rt_scratch(ctx, sizeof(double) * 6 * (size_t)nblk, &cbuf);  // leaf arena: free between the two calls
  if (rc != RT_OK) return rc;
  vp.C = static_cast<const double*>(cbuf);
  static const int eig_flags_v = [] { const char* e = getenv("ROMTIME_EIG_FLAGS"); return e ? atoi(e) : 0; }();
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&symeig_vectors_kernel<128>), 80 * 1024));
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&symeig_vectors_kernel<256>), 80 * 1024));
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&symeig_vectors_kernel<512>), 80 * 1024));
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&symeig_vectors_kernel<1024>), 80 * 1024));
  if (n > 512) {
    const size_t vlds = sizeof(double) * (8 * 1024 + 8) + 1024;
    hipLaunchKernelGGL(symeig_wy_kernel<1024>, dim3((unsigned)nblk), dim3(64), 0, ctx->stream, vp.V, (int)n,
                       static_cast<double*>(cbuf));
    hipLaunchKernelGGL(symeig_vectors_kernel<1024>, dim3((unsigned)k), dim3(64), vlds, ctx->stream, vp);
  } else if (n <= 128 && !(eig_flags_v & 2)) {
    const size_t vlds = sizeof(double) * (8 * 128 + 8) + 128;
    hipLaunchKernelGGL(symeig_wy_kernel<128>, dim3((unsigned)nblk), dim3(64), 0, ctx->stream, vp.V, (int)n,
                       static_cast<double*>(cbuf));
    hipLaunchKernelGGL(symeig_vectors_kernel<128>, dim3((unsigned)k), dim3(64), vlds, ctx->stream, vp);
  } else if (n <= 256 && !(eig_flags_v & 2)) {
    const size_t vlds = sizeof(double) * (8 * 256 + 8) + 256;
    hipLaunchKernelGGL(symeig_wy_kernel<256>, dim3((unsigned)nblk), dim3(64), 0, ctx->stream, vp.V, (int)n,
                       static_cast<double*>(cbuf));
    hipLaunchKernelGGL(symeig_vectors_kernel<256>, dim3((unsigned)k), dim3(64), vlds, ctx->stream, vp);
  } else {
    const size_t vlds = sizeof(double) * (8 * 512 + 8) + 512;
    hipLaunchKernelGGL(symeig_wy_kernel<512>, dim3((unsigned)nblk), dim3(64), 0, ctx->stream, vp.V, (int)n,
                       static_cast<double*>(cbuf));
    hipLaunchKernelGGL(symeig_vectors_kernel<512>, dim3((unsigned)k), dim3(64), vlds, ctx->stream, vp);
  }
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}
