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
#include "sweep_advance.h"
#include "tile_layout.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int SOLVE_THREADS = 256;
constexpr int SOLVE_WAVES = SOLVE_THREADS / 64;
constexpr int SOLVE_BATCH = 4;

// LU with partial pivoting of the r x r matrix A (LDS, leading dimension lda, overwritten) and the solve with the
// right-hand side b (LDS, overwritten); the solution goes to xout (global, r values).  NT threads, all of which must
// call it; `parts` threads share a row.  Returns 1 when a pivot was exactly zero.  The matrix and b must be in LDS
// and visible (barrier) on entry; xout is complete after the caller's next barrier.
// `perm_out` (LDS, r ints, optional): the pivot row of every column; with it the multipliers are kept in place of the
// eliminated entries (A[i][c] = l_ic for the rows i still live at column c), so that further right-hand sides can be
// solved against the factors (dense_solve_multi_kernel).
template <int NT>
__device__ __forceinline__ int lu_solve_lds(double* A, int lda, double* b, int r, int parts, double* rb,
                                            int* perm_out = nullptr) {
  constexpr int WAVES = NT / 64;
  __shared__ double s_pv[2][WAVES];
  __shared__ int s_pi[2][WAVES];
  __shared__ int s_perm[128];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int row = tid / parts, part = tid - row * parts;
  const bool owner = row < r;
  bool retired = !owner;
  double* Ar = A + (owner ? row : 0) * lda;
  int sing = 0;
  int cm = 1 % parts;  // (c + 1) mod parts
  double l_keep = 0.0;
  int c_keep = -1;

  double cand = (owner && part == 0) ? fabs(Ar[0]) : -1.0;  // candidates of column 0
  for (int c = 0; c < r; ++c) {
    int ci = row;
    rtw::wave_argmax(cand, ci);  // (|value| desc, row asc) over the wave's live rows
    if (lane == 0) {
      s_pv[c & 1][wid] = cand;
      s_pi[c & 1][wid] = ci;
    }
    __syncthreads();
    if (c_keep >= 0) {
      Ar[c_keep] = l_keep;
      c_keep = -1;
    }
    double best = s_pv[c & 1][0];
    int pr = s_pi[c & 1][0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) {
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
      if (perm_out && part == 0) {   // written after the NEXT barrier: the row's other threads are reading Ar[c] now
        l_keep = l;
        c_keep = c;
      }
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
  if (perm_out) {
    if (c_keep >= 0) Ar[c_keep] = l_keep;
    for (int c = tid; c < r; c += NT) perm_out[c] = s_perm[c];
    __syncthreads();
  }

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
  return sing;
}

__global__ __launch_bounds__(SOLVE_THREADS) void dense_solve_kernel(const double* __restrict__ K,
                                                                    double* __restrict__ rhs, int r, int parts,
                                                                    int* info, const int* only_if, long* counters,
                                                                    const rt_advance adv) {
  if (only_if && only_if[blockIdx.x] == 0) return;  // fallback launch: only the systems another solver gave up on
  if (only_if && threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[RT_CNT_LU_FALLBACK]), 1ull);
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lda = r | 1;  // odd leading dimension: column walks hit distinct banks
  double* A = sm;         // r x lda
  double* b = sm + (size_t)r * lda;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double* Kb = K + (size_t)blockIdx.x * r * r;
  double* rb = rhs + (size_t)blockIdx.x * r;
  for (int i = wid; i < r; i += SOLVE_WAVES)
    for (int j = lane; j < r; j += 64) A[i * lda + j] = Kb[i * r + j];
  for (int e = tid; e < r; e += SOLVE_THREADS) b[e] = rb[e];
  __syncthreads();
  const int sing = lu_solve_lds<SOLVE_THREADS>(A, lda, b, r, parts, rb);
  if (info && tid == 0) info[blockIdx.x] = sing ? RT_WARN_SINGULAR : 0;
  if (adv.enabled) {  // the hyper-reduced sweep's end of step for this system (its solver kernel left it alone)
    __syncthreads();  // rb is complete and visible in this workgroup
    hsweep_advance_rows(adv, blockIdx.x, r, rhs, 1, b, tid, SOLVE_THREADS);
  }
}

// K X = B for MANY right-hand sides against one matrix (folding an interpolation matrix PT_U into the expansion of a
// hyper-reduced operator, Z = basis_rom PT_U^-1: np.linalg.solve with r^2 right-hand sides in the reference's terms,
// deim.py:477-493 applied once to every column).  Every workgroup factorises K for itself in LDS (pivoted LU with the
// multipliers kept) and then gives each of its threads one right-hand side: forward and back substitution in pivot
// order, the column's intermediate values in the output array (coalesced across the threads: the right-hand-side index
// runs fastest), the factors broadcast out of LDS.
__global__ __launch_bounds__(SOLVE_THREADS) void dense_solve_multi_kernel(const double* __restrict__ K, int r, int parts,
                                                                          const double* __restrict__ B,
                                                                          double* __restrict__ X, long nrhs, int* info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lda = r | 1;
  double* A = sm;                                    // r x lda
  double* dummy = sm + (size_t)r * lda;              // r: the single right-hand side lu_solve_lds carries along
  int* perm = reinterpret_cast<int*>(dummy + r);     // r
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = wid; i < r; i += SOLVE_WAVES)
    for (int j = lane; j < r; j += 64) A[i * lda + j] = K[(size_t)i * r + j];
  for (int e = tid; e < r; e += SOLVE_THREADS) dummy[e] = 0.0;
  __syncthreads();
  const int sing = lu_solve_lds<SOLVE_THREADS>(A, lda, dummy, r, parts, dummy, perm);   // (its own solve goes to LDS, unused)
  __syncthreads();
  if (info && tid == 0 && blockIdx.x == 0) info[0] = sing ? RT_WARN_SINGULAR : 0;
  const long col = (long)blockIdx.x * SOLVE_THREADS + tid;
  if (col >= nrhs) return;
  // forward: y_c = b[p_c] - sum_{k<c} L[p_c][k] y_k   (row p_c was live at every column k < c: its multipliers sit there)
  for (int c = 0; c < r; ++c) {
    const double* row = A + perm[c] * lda;
    double acc = B[(size_t)perm[c] * nrhs + col];
    for (int k = 0; k < c; ++k) acc = fma(-row[k], X[(size_t)k * nrhs + col], acc);
    X[(size_t)c * nrhs + col] = acc;
  }
  // back: x_c = (y_c - sum_{k>c} U[p_c][k] x_k) / U[p_c][c]
  for (int c = r - 1; c >= 0; --c) {
    const double* row = A + perm[c] * lda;
    double acc = X[(size_t)c * nrhs + col];
    for (int k = c + 1; k < r; ++k) acc = fma(-row[k], X[(size_t)k * nrhs + col], acc);
    X[(size_t)c * nrhs + col] = acc / row[c];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Inverse tracking for the online sweep.  Consecutive time steps solve with matrices that differ by O(dt), and
// pivoted LU of an 80 x 80 system is a chain of ~80 x 3500 dependent cycles no matter how many threads help.
// Newton-Schulz  X <- X (2 I - K X)  converges quadratically to K^-1 from the previous step's inverse
// (||I - K X|| ~ 1e-4 -> 1e-8 -> 1e-16) and is two small GEMMs on the matrix cores per iteration, all operands
// resident in LDS.  One workgroup per system: iterate until the residual before the last update was < 1e-6
// (so < 1e-12 after it), then x = X b with one step of iterative refinement against K itself, which makes the
// answer as accurate as the direct solve.  Without a usable start (first step, or ||I - K X|| > 0.7) it
// restarts from X = K^T / (||K||_1 ||K||_inf), which always converges for a nonsingular K.
constexpr int NS_THREADS = 512;
constexpr int NS_MAX_ITER = 100;
constexpr int NS_REFINE_MAX = 8;         // refinement steps tried with the carried inverse before Newton-Schulz takes over
constexpr int NS_REFRESH_AFTER = 5;      // more steps than this: the answer stands, and X is refreshed for the steps to come
constexpr double NS_REFINE_TOL = 2e-15;  // ||b - K x|| <= tol ||b||: the floor of a refinement in working precision
constexpr double NS_REFINE_RATE = 0.3;   // a step must shrink the residual at least this much, or X is not worth keeping

__device__ __forceinline__ double ns_block_sum(double x, double* s_red, int tid) {
  x = rtw::wave_sum(x);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = x;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < NS_THREADS / 64; ++w) t += s_red[w];
  return t;
}

__device__ __forceinline__ double ns_block_max(double x, double* s_red, int tid) {
  x = rtw::wave_max(x);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = x;
  __syncthreads();
  double t = s_red[0];
#pragma unroll
  for (int w = 1; w < NS_THREADS / 64; ++w) t = fmax(t, s_red[w]);
  return t;
}

__global__ __launch_bounds__(NS_THREADS) void newton_solve_kernel(const double* __restrict__ K,
                                                                  double* __restrict__ Xinv,
                                                                  double* __restrict__ rhs, int r, int S,
                                                                  int have_prev, int* __restrict__ info,
                                                                  const rt_newton_rhs rq, long* counters,
                                                                  const rt_advance adv) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tr = (r + 15) / 16, rp = tr * 16;
  double* sK = sm;                    // [rp][S], padded with the identity
  double* sX = sK + (size_t)rp * S;   // [rp][S]
  double* sT = sX + (size_t)rp * S;   // [rp][S]
  __shared__ double s_red[NS_THREADS / 64];
  __shared__ double s_vec[4][96];   // r <= 80 here (three padded r x r matrices in the dynamic part)
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* Kb = K + (size_t)blockIdx.x * r * r;
  double* Xb = Xinv + (size_t)blockIdx.x * r * r;
  double* rb = rhs + (size_t)blockIdx.x * r;

  // out_i = sum_j A[i][j] v[j] for the r rows of an LDS matrix (A, v, out in LDS; out may not be v): eight lanes
  // share a row (columns j == lane mod 8) and add up with three butterfly steps, 64 rows per pass of the workgroup.
  // No LDS scratch: the three padded matrices leave 3 KB of the CU's 160.  (One thread per row walked r LDS round
  // trips one after the other, four times per solve.)
  auto matvec = [&](const double* A, const double* v, double* out) {
    const int sub = lane & 7;
    for (int i0 = 0; i0 < r; i0 += NS_THREADS / 8) {
      const int i = i0 + (tid >> 3);
      double acc = 0.0;
      if (i < r) {
        const double* row = A + i * S;
        for (int j = sub; j < r; j += 8) acc = fma(row[j], v[j], acc);
      }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      acc += __shfl_xor(acc, 4);
      if (i < r && sub == 0) out[i] = acc;
    }
    __syncthreads();
  };

  // K, the carried inverse and (when the right-hand side is formed here) M_N come in together: one round of global
  // loads, M_N parked in sT until the first product needs the space
  const double* Mb = rq.MN ? rq.MN + (size_t)blockIdx.x * (rq.mn_stride < 0 ? (long)r * r : rq.mn_stride) : nullptr;
  const bool wide_ok = (r == rp) && ((reinterpret_cast<size_t>(Kb) | reinterpret_cast<size_t>(Xb) | reinterpret_cast<size_t>(Mb)) & 15) == 0;
  if (wide_ok) {
    // r a multiple of 16: a row is r / 2 <= 40 pairs, one 16-byte load per lane and row; the <= 10 rows of a wave
    // for all three matrices are in flight together
    constexpr int QMAX = 10;
    d2 kv[QMAX], xv[QMAX], mv[QMAX];
    const int j = 2 * lane;
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
      const int i = wid + 8 * q;
      const bool in = i < r && j < r;
      kv[q] = in ? *reinterpret_cast<const d2*>(Kb + i * r + j) : d2{0.0, 0.0};
      xv[q] = (in && have_prev) ? *reinterpret_cast<const d2*>(Xb + i * r + j) : d2{i == j ? 1.0 : 0.0, i == j + 1 ? 1.0 : 0.0};
      mv[q] = (in && Mb) ? *reinterpret_cast<const d2*>(Mb + i * r + j) : d2{0.0, 0.0};
    }
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
      const int i = wid + 8 * q;
      if (i < r && j < r) {
        *reinterpret_cast<d2*>(sK + i * S + j) = kv[q];
        *reinterpret_cast<d2*>(sX + i * S + j) = xv[q];
        if (Mb) *reinterpret_cast<d2*>(sT + i * S + j) = mv[q];
      }
    }
  } else {
    // rp <= 80: a wave has at most 10 rows (i = wid + 8 q) of at most 2 x 64 columns.  Fixed trip counts, loads of
    // half the rows issued before any LDS store: a loop over runtime bounds went load - wait - store twenty times
    constexpr int QH = 5;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      double kv[QH][2], xv[QH][2], mv[QH][2];
#pragma unroll
      for (int q = 0; q < QH; ++q)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int i = wid + 8 * (QH * half + q), j = lane + 64 * c;
          const bool in = i < r && j < r;
          const double eye = (i == j) ? 1.0 : 0.0;
          kv[q][c] = in ? Kb[i * r + j] : eye;
          xv[q][c] = (in && have_prev) ? Xb[i * r + j] : eye;
          mv[q][c] = (in && Mb) ? Mb[i * r + j] : 0.0;
        }
#pragma unroll
      for (int q = 0; q < QH; ++q)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int i = wid + 8 * (QH * half + q), j = lane + 64 * c;
          if (i < rp && j < rp) {
            sK[i * S + j] = kv[q][c];
            sX[i * S + j] = xv[q][c];
            if (Mb) sT[i * S + j] = mv[q][c];
          }
        }
    }
  }
  if (rq.MN) {
    // right-hand side: b = M_N (c0 u^n + c1 u^{n-1}) + dt Zf^T F_rhs (also left in rhs for the fallback); row i is
    // thread i's: r FMAs out of LDS and mf independent loads - a wave per row with a cross-lane sum took a chain
    // of r / 8 global round trips per wave
    if (tid < r) s_vec[1][tid] = rq.c0 * rq.un[(size_t)blockIdx.x * r + tid] + rq.c1 * rq.unm1[(size_t)blockIdx.x * r + tid];
    __syncthreads();
    double f = 0.0;
    if (tid < r) {
      const double* Ff = (rq.ctr ? rq.Ff + *rq.ctr * rq.ff_stride : rq.Ff) + (size_t)blockIdx.x * rq.mf;
      for (int e = 0; e < rq.mf; ++e) f = fma(Ff[e], rq.Zf[(size_t)e * r + tid], f);
    }
    matvec(sT, s_vec[1], s_vec[2]);
    if (tid < r) {
      const double v = fma(rq.dt, f, s_vec[2][tid]);
      s_vec[0][tid] = v;
      rb[tid] = v;
    }
  } else if (tid < r) {
    s_vec[0][tid] = rb[tid];
  }
  __syncthreads();

  // C(ti, tj) = sum_k A[16 ti + i][k] B[k][16 tj + j] on the matrix cores, operands out of LDS.  A wave owns one
  // rectangular block of tiles (tile_layout.h; at most 2 x 3 for r <= 80), so a k-step costs ni + nj LDS reads for
  // ni nj MFMAs (scattered tiles t = wid + 8 q read two operands per MFMA and took 9 us per 80^3 product).
  constexpr int NBI = 2, NBJ = 3;
  const Blk bk = tile_block(tr, wid);
  auto tiles_product = [&](const double* A, const double* Bm, d4 (&acc)[NBI][NBJ]) {
    const double* ap[NBI];
    const double* bp[NBJ];
#pragma unroll
    for (int i = 0; i < NBI; ++i) ap[i] = A + (16 * (bk.i0 + (i < bk.ni ? i : 0)) + l15) * S + l4;
#pragma unroll
    for (int j = 0; j < NBJ; ++j) bp[j] = Bm + l4 * S + 16 * (bk.j0 + (j < bk.nj ? j : 0)) + l15;
#pragma unroll
    for (int i = 0; i < NBI; ++i)
#pragma unroll
      for (int j = 0; j < NBJ; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    for (int k16 = 0; k16 < rp / 16; ++k16)   // rp is a multiple of 16: four k-steps per trip, unrolled
#pragma unroll
    for (int ku = 0; ku < 4; ++ku) {
      const int k4 = 4 * k16 + ku;
      double a[NBI], bq[NBJ];
#pragma unroll
      for (int i = 0; i < NBI; ++i) a[i] = ap[i][4 * k4];
#pragma unroll
      for (int j = 0; j < NBJ; ++j) bq[j] = bp[j][4 * k4 * S];
#pragma unroll
      for (int i = 0; i < NBI; ++i)
#pragma unroll
        for (int j = 0; j < NBJ; ++j)
          if (i < bk.ni && j < bk.nj)  // wave-uniform
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bq[j], acc[i][j], 0, 0, 0);
    }
  };

  auto restart = [&]() {  // X = K^T / (||K||_1 ||K||_inf)
    double rs = 0.0, cs = 0.0;
    if (tid < r) {
      for (int j = 0; j < r; ++j) {
        rs += fabs(sK[tid * S + j]);
        cs += fabs(sK[j * S + tid]);
      }
    }
    const double ninf = ns_block_max(rs, s_red, tid);
    const double n1 = ns_block_max(cs, s_red, tid);
    const double sc = 1.0 / (n1 * ninf);
    for (int i = wid; i < rp; i += NS_THREADS / 64)
      for (int j = lane; j < rp; j += 64) sX[i * S + j] = (i < r && j < r) ? sK[j * S + i] * sc : ((i == j) ? 1.0 : 0.0);
    __syncthreads();
  };
  if (!have_prev) restart();

  // Round 3 - refinement before refreshment.  The carried inverse is a PRECONDITIONER: x <- x + X (b - K x) contracts
  // the error by ||I - X K|| per step, two r x r mat-vecs each, against the two r^3 products of a Newton-Schulz update.
  // X was exact to ~1e-8 when it was last refreshed and K has moved by O(dt) per time step since, so a few refinement
  // steps reach the residual floor for many time steps in a row; only when they take more than NS_REFRESH_AFTER steps
  // (or do not contract) is X refreshed - and in the first case the answer is already there, the refresh is for the
  // steps to come.  31 -> ~15 us per step for 32 systems of 80 (bench.py, hyper-reduced sweep).
  bool solved = false, refresh = true;
  if (have_prev) {
    double bpart = (tid < r) ? s_vec[0][tid] * s_vec[0][tid] : 0.0;
    const double bnorm2 = ns_block_sum(bpart, s_red, tid);
    matvec(sX, s_vec[0], s_vec[1]);                        // x = X b
    double prev = 1e300;
    int used = 0;
    for (int it = 0; it <= NS_REFINE_MAX; ++it) {
      matvec(sK, s_vec[1], s_vec[2]);                      // K x
      double rr = 0.0;
      if (tid < r) {
        const double e = s_vec[0][tid] - s_vec[2][tid];
        s_vec[2][tid] = e;
        rr = e * e;
      }
      const double rn2 = ns_block_sum(rr, s_red, tid);     // barriers inside: the residual vector is complete
      if (!(rn2 == rn2)) break;                             // NaN: leave it to the tracked route and its fallbacks
      if (rn2 <= NS_REFINE_TOL * NS_REFINE_TOL * bnorm2 || (it > 0 && rn2 >= 0.25 * prev && rn2 <= 1e-26 * bnorm2)) {
        solved = true;                                      // at the floor (or within a hair of it and no longer moving)
        used = it;
        break;
      }
      if (it == NS_REFINE_MAX || rn2 > NS_REFINE_RATE * NS_REFINE_RATE * prev) break;   // out of steps / no contraction
      prev = rn2;
      matvec(sX, s_vec[2], s_vec[3]);                      // X r
      if (tid < r) s_vec[1][tid] += s_vec[3][tid];
      __syncthreads();
    }
    if (solved) {
      refresh = used > NS_REFRESH_AFTER;
      if (tid < r) rb[tid] = s_vec[1][tid];
      if (!refresh) {                                       // X stays as it is: nothing to write back
        if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[RT_CNT_SOLVES]), 1ull);
        if (info && tid == 0) info[blockIdx.x] = 0;
        if (adv.enabled) {
          __syncthreads();
          hsweep_advance_rows(adv, blockIdx.x, r, rhs, 1, s_vec[1], tid, NS_THREADS);
        }
        return;
      }
    }
  }

  int status = RT_WARN_SINGULAR;
  bool restarted = !have_prev;
  int n_iter = 0, n_restart = 0;
  for (int it = 0; it < NS_MAX_ITER; ++it) {
    ++n_iter;
    // T = K X and the residual ||I - T||_F
    double part = 0.0;
    d4 acc[NBI][NBJ];
    tiles_product(sK, sX, acc);
#pragma unroll
    for (int i = 0; i < NBI; ++i)
#pragma unroll
      for (int j = 0; j < NBJ; ++j)
        if (i < bk.ni && j < bk.nj) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int row = 16 * (bk.i0 + i) + l4 + 4 * c, col = 16 * (bk.j0 + j) + l15;
            sT[row * S + col] = acc[i][j][c];
            const double e = ((row == col) ? 1.0 : 0.0) - acc[i][j][c];
            part = fma(e, e, part);
          }
        }
    const double res = sqrt(ns_block_sum(part, s_red, tid));  // barriers inside: sT complete
    if (res != res) break;  // NaN: singular to working precision (or bad input)
    if (!restarted && !(res < 0.7)) {  // the carried inverse is no contraction for this K: safe start instead
      restarted = true;
      n_restart = 1;
      restart();
      continue;
    }
    // X <- 2 X - X T
    tiles_product(sX, sT, acc);
    __syncthreads();  // every wave has finished reading the old X
#pragma unroll
    for (int i = 0; i < NBI; ++i)
#pragma unroll
      for (int j = 0; j < NBJ; ++j)
        if (i < bk.ni && j < bk.nj) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int row = 16 * (bk.i0 + i) + l4 + 4 * c, col = 16 * (bk.j0 + j) + l15;
            sX[row * S + col] = 2.0 * sX[row * S + col] - acc[i][j][c];
          }
        }
    __syncthreads();
    if (res < 1e-6) {
      status = 0;
      break;
    }
  }

  if (tid == 0) {  // rt_last_sweep_stats
    atomicAdd(reinterpret_cast<unsigned long long*>(&counters[RT_CNT_NS_ITER]), (unsigned long long)n_iter);
    atomicAdd(reinterpret_cast<unsigned long long*>(&counters[RT_CNT_NS_RESTART]), (unsigned long long)n_restart);
    atomicAdd(reinterpret_cast<unsigned long long*>(&counters[RT_CNT_SOLVES]), 1ull);
  }
  if (status != 0 && solved) {
    // the refinement had already delivered the answer; only the refresh of X failed: start afresh next time
    for (int i = wid; i < r; i += NS_THREADS / 64)
      for (int j = lane; j < r; j += 64) Xb[i * r + j] = 0.0;
    if (info && tid == 0) info[blockIdx.x] = 0;
    if (adv.enabled) {
      __syncthreads();
      hsweep_advance_rows(adv, blockIdx.x, r, rhs, 1, s_vec[1], tid, NS_THREADS);
    }
    return;
  }
  if (status != 0) {
    // The tracking gave up on this system (singular to working precision, or no contraction even from the safe
    // start): pivoted LU right here, out of the copy of K this workgroup already holds, and the next call starts
    // afresh.  (It used to be a second launch per step that found nothing to do in all but a handful of steps.)
    for (int i = wid; i < r; i += NS_THREADS / 64)
      for (int j = lane; j < r; j += 64) Xb[i * r + j] = 0.0;
    if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[RT_CNT_LU_FALLBACK]), 1ull);
    int lu_parts = NS_THREADS / r;
    if (lu_parts > 8) lu_parts = 8;
    __syncthreads();
    const int sing = lu_solve_lds<NS_THREADS>(sK, S, s_vec[0], r, lu_parts, rb);
    if (info && tid == 0) info[blockIdx.x] = sing ? RT_WARN_SINGULAR : 0;
    if (adv.enabled) {
      __syncthreads();
      hsweep_advance_rows(adv, blockIdx.x, r, rhs, 1, s_vec[1], tid, NS_THREADS);
    }
    return;
  }
  if (!solved) {   // x = X b, one refinement step against K:  x += X (b - K x)
    matvec(sX, s_vec[0], s_vec[1]);
    const double x = (tid < r) ? s_vec[1][tid] : 0.0;
    matvec(sK, s_vec[1], s_vec[2]);
    if (tid < r) s_vec[2][tid] = s_vec[0][tid] - s_vec[2][tid];
    __syncthreads();
    matvec(sX, s_vec[2], s_vec[1]);
    if (tid < r) rb[tid] = x + s_vec[1][tid];
  }
  if (wide_ok) {   // the refreshed inverse goes back the way it came: one 16-byte store per lane and row
    const int j = 2 * lane;
    d2 xv[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const int i = wid + 8 * q;
      if (i < r && j < r) xv[q] = *reinterpret_cast<const d2*>(sX + i * S + j);
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const int i = wid + 8 * q;
      if (i < r && j < r) *reinterpret_cast<d2*>(Xb + i * r + j) = xv[q];
    }
  } else {
    for (int i = wid; i < r; i += NS_THREADS / 64)
      for (int j = lane; j < r; j += 64) Xb[i * r + j] = sX[i * S + j];
  }
  if (info && tid == 0) info[blockIdx.x] = status;
  if (adv.enabled) {  // the hyper-reduced sweep's end of step for this system
    __syncthreads();  // rb is complete and visible in this workgroup
    hsweep_advance_rows(adv, blockIdx.x, r, rhs, 1, s_vec[1], tid, NS_THREADS);
  }
}

}  // namespace


static int dense_solve_launch(rt_ctx* ctx, double* K, double* rhs, int64_t r, int64_t B, int* info, const int* only_if,
                              const rt_advance* advance);

extern "C" int rt_dense_solve_batched(rt_ctx* ctx, double* K, double* rhs, int64_t r, int64_t B, int* info) {
  return dense_solve_launch(ctx, K, rhs, r, B, info, nullptr, nullptr);
}

static int dense_solve_launch(rt_ctx* ctx, double* K, double* rhs, int64_t r, int64_t B, int* info, const int* only_if,
                              const rt_advance* advance) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, K && rhs && r >= 1 && B >= 1);
  if (r > 128) {
    ctx->err = "rt_dense_solve_batched: r > 128 not supported (matrix must fit the CU's LDS)";
    return RT_ERR_UNSUPPORTED;
  }
  const int lda = (int)r | 1;
  const size_t lds = sizeof(double) * ((size_t)r * lda + r);
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&dense_solve_kernel), 140 * 1024));
  int parts = SOLVE_THREADS / (int)r;  // threads per row
  if (parts > 8) parts = 8;
  hipLaunchKernelGGL(dense_solve_kernel, dim3((unsigned)B), dim3(SOLVE_THREADS), lds, ctx->stream, K, rhs, (int)r,
                     parts, info, only_if, ctx->dev_counters, advance ? *advance : rt_advance{});
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

extern "C" int rt_dense_solve_multi(rt_ctx* ctx, const double* K, int64_t r, const double* B, double* X, int64_t nrhs,
                                   int* info) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, K && B && X && B != X && r >= 1 && nrhs >= 1);
  if (r > 128) {
    ctx->err = "rt_dense_solve_multi: r > 128 not supported (the factors must fit the CU's LDS)";
    return RT_ERR_UNSUPPORTED;
  }
  const int lda = (int)r | 1;
  const size_t lds = sizeof(double) * ((size_t)r * lda + r) + sizeof(int) * (size_t)r;
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&dense_solve_multi_kernel), 140 * 1024));
  int parts = SOLVE_THREADS / (int)r;
  if (parts > 8) parts = 8;
  const unsigned grid = (unsigned)((nrhs + SOLVE_THREADS - 1) / SOLVE_THREADS);
  hipLaunchKernelGGL(dense_solve_multi_kernel, dim3(grid), dim3(SOLVE_THREADS), lds, ctx->stream, K, (int)r, parts, B, X,
                     (long)nrhs, info);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

// Internal (sweep.hip): solve K_b x_b = rhs_b while tracking K_b^-1 in Xinv (B x r x r, caller-owned, carried
// from call to call).  have_prev = 0 on the first call.  RT_ERR_UNSUPPORTED when three padded r x r matrices
// do not fit the LDS (r > 80): the caller uses rt_dense_solve_batched.  With `recipe` the kernel forms the
// right-hand side itself (and leaves it in rhs) instead of reading it.
int rt_newton_solve_batched(rt_ctx* ctx, const double* K, double* Xinv, double* rhs, int64_t r, int64_t B,
                            int have_prev, int* info, const rt_newton_rhs* recipe, const rt_advance* advance) {
  const int rp = (int)((r + 15) / 16) * 16;
  int S = rp;
  while (S % 4 != 2) ++S;  // 2 S == 4 (mod 8): the 16 rows of an A-operand read fall in distinct LDS banks
  const size_t lds = sizeof(double) * 3 * (size_t)rp * S;
  if (lds > 155 * 1024) return RT_ERR_UNSUPPORTED;  // + 3.1 KB of static LDS <= the CU's 160 KB
  RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&newton_solve_kernel), 155 * 1024));
  rt_newton_rhs rq{};
  if (recipe) rq = *recipe;
  hipLaunchKernelGGL(newton_solve_kernel, dim3((unsigned)B), dim3(NS_THREADS), lds, ctx->stream, K, Xinv, rhs, (int)r, S,
                     have_prev, info, rq, ctx->dev_counters, advance ? *advance : rt_advance{});
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

extern "C" int rt_tracked_solve_batched(rt_ctx* ctx, const double* K, double* Xinv, double* rhs, int64_t r, int64_t B,
                                        int have_prev, int* info) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, K && Xinv && rhs && r >= 1 && B >= 1);
  return rt_newton_solve_batched(ctx, K, Xinv, rhs, r, B, have_prev ? 1 : 0, info, nullptr, nullptr);
}
