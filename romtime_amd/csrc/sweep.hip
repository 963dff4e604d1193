// Online reduced sweep on the device (RomConstructor*.solve, rom.py:430-555 with the direct-path
// assemble_system / assemble_system_rhs of rom.py:877-929), for n_mu parameter points at once.
//
// Per time step, all on the ctx stream, nothing returns to the host:
//   k[b]   = bdf M + dt (sum_q theta_q(mu_b, t) A_q + diag(u*_b) T)           value vectors of the step (one pass)
//   K_N[b] = V^T (A[k[b]] V)                                                  fused SpMM + MFMA projection
//   b_N[b] = M_N (2 u^n - u^{n-1}/2) + dt sum_f phi_f(mu_b, t) (V^T f_f)     (BDF2; M_N u^n + ... for BDF1 / step 0)
//   u^{n+1}[b] = K_N[b]^-1 b_N[b]                                            batched pivoted LU in LDS
//   u_h[b] = V u^{n+1}[b],   u* = 2 u_h - u_h^{n-1}                          next step's trilinear state
// The operators are affine in precomputed value vectors on one CSR pattern (the form MDEIM itself
// produces, and what a closed-form 1-D assembly yields); the state-dependent term is diag(u*) T.
#include <cstdlib>

#include "common.h"
#include "sweep_advance.h"

typedef double xd2 __attribute__((ext_vector_type(2)));
typedef double xd4 __attribute__((ext_vector_type(4)));

namespace {

// The expansion of the hyper-reduced step,  C (R x N) = G (R x K) Z (K x N)  with R <= 64 rows (K_N and M_N of all
// parameter points), a SHORT contraction (K = the interpolation coefficients of all operators, a few hundred) and a
// long N (r^2).  The generic GEMM walks the contraction in 16-deep stages with one stage of look-ahead: 18 stages of
// ~0.9 us each, every one a global-memory round trip that 8 MFMAs per wave cannot hide - 16 us per step.  Here a
// workgroup owns a 64 x 32 tile and takes the contraction in phases of EX_KP = 144: ALL loads of a phase (G's 64 rows,
// the phase's 144 rows of Z's 32 columns) are issued at once, so a phase pays the memory latency once; two phases for
// K = 280.  Operands through LDS in the two conflict-free images of the other kernels; each wave 16 rows x 32 columns.
constexpr int EX_KP = 144, EX_SA = EX_KP + 2, EX_SB = 34, EX_THREADS = 256;
constexpr int EX_LA = 64 * EX_KP / 2 / EX_THREADS;   // d2 loads of G per thread and phase (18)
constexpr int EX_LB = EX_KP * 32 / 2 / EX_THREADS;   // d2 loads of Z per thread and phase (9)

template <int MODE>   // 0; timing ablations (ROMTIME_SWEEP_FLAGS, results wrong): 6 = no global loads, 8 = one k-step, 14 = both
__global__ __launch_bounds__(EX_THREADS) void expansion_kernel(const double* __restrict__ G, long ldg,
                                                                const double* __restrict__ Z, long ldz,
                                                                double* __restrict__ Cm, long ldc, int R, int K, long N) {
  constexpr int mode = MODE;
  extern __shared__ __attribute__((aligned(16))) double ex_sm[];
  double* sA = ex_sm;                    // [64][EX_SA]: G rows, contraction contiguous
  double* sB = ex_sm + 64 * EX_SA;       // [EX_KP][EX_SB]: Z rows of this tile's 32 columns
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long n0 = (long)blockIdx.x * 32;
  // four independent accumulator chains per wave (two column tiles x the parity of the k-step): with one wave per SIMD
  // nothing else hides the latency of an MFMA that waits for the previous one into the same accumulator
  xd4 acc[2][2] = {{xd4{0, 0, 0, 0}, xd4{0, 0, 0, 0}}, {xd4{0, 0, 0, 0}, xd4{0, 0, 0, 0}}};
  xd2 ra[EX_LA], rb[EX_LB];
  // Branch-free loads: every load goes to a clamped, aligned address and is zeroed afterwards where it was out of range
  // (even K, N and leading dimensions: the host checks) - a predicate per load makes the compiler wait for each load
  // where it is issued.
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < EX_LA; ++i) {
      const int q = tid + EX_THREADS * i, row = q / (EX_KP / 2), kk = 2 * (q % (EX_KP / 2));
      const int k = k0 + kk;
      const bool ok = row < R && k < K;
      const long off = (long)(row < R ? row : R - 1) * ldg + (k < K ? k : K - 2);
      const xd2 v = (mode & 2) ? xd2{1.0, 1.0} : *reinterpret_cast<const xd2*>(G + off);
      ra[i] = ok ? v : xd2{0.0, 0.0};
    }
#pragma unroll
    for (int i = 0; i < EX_LB; ++i) {
      const int q = tid + EX_THREADS * i, kk = q / 16, j = 2 * (q % 16);
      const int k = k0 + kk;
      const bool ok = k < K && n0 + j < N;
      const long off = (long)(k < K ? k : K - 1) * ldz + (n0 + j < N ? n0 + j : N - 2);
      const xd2 v = (mode & 4) ? xd2{1.0, 1.0} : *reinterpret_cast<const xd2*>(Z + off);
      rb[i] = ok ? v : xd2{0.0, 0.0};
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < EX_LA; ++i) {
      const int q = tid + EX_THREADS * i, row = q / (EX_KP / 2), kk = 2 * (q % (EX_KP / 2));
      *reinterpret_cast<xd2*>(&sA[row * EX_SA + kk]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < EX_LB; ++i) {
      const int q = tid + EX_THREADS * i, kk = q / 16, j = 2 * (q % 16);
      *reinterpret_cast<xd2*>(&sB[kk * EX_SB + j]) = rb[i];
    }
  };
  const double* fa = sA + (16 * wid + l15) * EX_SA + l4;   // A operand: row 16 w + l15, k = 4 k4 + l4
  const double* fb = sB + l4 * EX_SB + l15;                 // B operand: k = 4 k4 + l4, column 16 j + l15
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += EX_KP) {
    commit();
    __syncthreads();
    if (k0 + EX_KP < K) fetch(k0 + EX_KP);   // the next phase's loads fly while this one is multiplied
#pragma unroll
    for (int k4 = 0; k4 < ((mode & 8) ? 2 : EX_KP / 4); ++k4) {
      const double a = fa[4 * k4];
      const double b0 = fb[4 * k4 * EX_SB], b1 = fb[4 * k4 * EX_SB + 16];
      acc[0][k4 & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[0][k4 & 1], 0, 0, 0);
      acc[1][k4 & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[1][k4 & 1], 0, 0, 0);
    }
    __syncthreads();   // the operands have been consumed: the next phase may overwrite them
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = 16 * wid + l4 + 4 * c;
      const long col = n0 + 16 * j + l15;
      if (row < R && col < N) Cm[(long)row * ldc + col] = acc[j][0][c] + acc[j][1][c];
    }
}

}  // namespace

// C = G Z for R <= 64 rows; RT_ERR_UNSUPPORTED beyond (the caller takes the generic GEMM).
int rt_expansion_gemm(rt_ctx* ctx, const double* G, long ldg, const double* Z, long ldz, double* Cm, long ldc, long R,
                      long K, long N) {
  static const int flags = [] { const char* e = getenv("ROMTIME_SWEEP_FLAGS"); return e ? atoi(e) : 0; }();
  const bool off = flags & 1;
  if (off || R > 64 || R < 1 || K < 2 || N < 2 || ((K | N | ldg | ldz) & 1) ||
      ((reinterpret_cast<size_t>(G) | reinterpret_cast<size_t>(Z)) & 15))
    return RT_ERR_UNSUPPORTED;
  const size_t lds = sizeof(double) * (64 * EX_SA + EX_KP * EX_SB);
  const dim3 grid((unsigned)((N + 31) / 32));
#define EX_LAUNCH(M_)                                                                                              \
  {                                                                                                                \
    RT_TRY(rt_func_lds(ctx, reinterpret_cast<const void*>(&expansion_kernel<M_>), (int)lds));                      \
    hipLaunchKernelGGL(expansion_kernel<M_>, grid, dim3(EX_THREADS), lds, ctx->stream, G, ldg, Z, ldz, Cm, ldc, (int)R, \
                       (int)K, N);                                                                                 \
  }
  switch (flags & 14) {
    case 6: EX_LAUNCH(6) break;
    case 8: EX_LAUNCH(8) break;
    case 14: EX_LAUNCH(14) break;
    default: EX_LAUNCH(0) break;
  }
#undef EX_LAUNCH
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

namespace {

__global__ void sweep_rows_kernel(const long* __restrict__ indptr, long N, int* __restrict__ row_of) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N) return;
  for (long e = indptr[row]; e < indptr[row + 1]; ++e) row_of[e] = (int)row;
}

// kval[b][e] = bdf*mass[e] + dt*(sum_q coef[b][q]*terms[q][e] + u*(b, row_of[e])*tril[e]),  u* = 2 uh - uh_prev
// One thread per entry e for ALL parameter points: the Q + 2 operator arrays are read once (not once per parameter
// point through the L2: 132 -> ~55 us at 5e5 entries x 32 points), the writes stream out vector by vector.
// COEF_LDS: the step's coefficient table coef[B][Q] is staged in LDS (the common case: 32 points x a handful of terms);
// a table beyond SV_LDS_BYTES (e.g. 64 points x 200 terms) is read through the L1/L2 instead - any B and Q work.
constexpr int SV_QMAX = 8;
constexpr long SV_LDS_BYTES = 32 * 1024;
template <bool COEF_LDS>
__global__ __launch_bounds__(256) void sweep_values_kernel(const double* __restrict__ mass,
                                                           const double* __restrict__ terms, int Q,
                                                           const double* __restrict__ coef,
                                                           const double* __restrict__ tril,
                                                           const int* __restrict__ row_of,
                                                           const double* __restrict__ uh,
                                                           const double* __restrict__ uhp, int extrapolate, long nnz,
                                                           long N, int B, double bdf, double dt,
                                                           double* __restrict__ kval) {
  extern __shared__ double s_coef[];   // [B][Q]
  if (COEF_LDS) {
    for (int i = threadIdx.x; i < B * Q; i += blockDim.x) s_coef[i] = coef[i];
    __syncthreads();
  }
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  double tq[SV_QMAX];
#pragma unroll
  for (int q = 0; q < SV_QMAX; ++q) tq[q] = (q < Q) ? terms[(long)q * nnz + e] : 0.0;
  const double m = bdf * mass[e];
  const double tv = tril ? tril[e] : 0.0;
  const long row = tril ? row_of[e] : 0;
#pragma unroll 4
  for (int b = 0; b < B; ++b) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < SV_QMAX; ++q)
      if (q < Q) s = fma(COEF_LDS ? s_coef[b * Q + q] : coef[b * Q + q], tq[q], s);
    for (int q = SV_QMAX; q < Q; ++q)                                             // more terms than registers
      s = fma(COEF_LDS ? s_coef[b * Q + q] : coef[b * Q + q], terms[(long)q * nnz + e], s);
    if (tril) {
      const double u = uh[(long)b * N + row];
      const double us = extrapolate ? 2.0 * u - uhp[(long)b * N + row] : u;
      s = fma(us, tv, s);
    }
    kval[(long)b * nnz + e] = fma(dt, s, m);
  }
}

// rhs[b] = M_N (c0 u_n[b] + c1 u_nm1[b]) + dt * sum_f rcoef[b][f] * fN[f]
__global__ void sweep_rhs_kernel(const double* __restrict__ MN, const double* __restrict__ un,
                                 const double* __restrict__ unm1, double c0, double c1, double dt,
                                 const double* __restrict__ rcoef, const double* __restrict__ fN, int F, int r,
                                 double* __restrict__ rhs) {
  extern __shared__ double su[];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < r) su[t] = c0 * un[(long)b * r + t] + c1 * unm1[(long)b * r + t];
  __syncthreads();
  if (t < r) {
    double acc = 0.0;
    for (int j = 0; j < r; ++j) acc = fma(MN[(long)t * r + j], su[j], acc);
    double f = 0.0;
    for (int q = 0; q < F; ++q) f = fma(rcoef[(long)b * F + q], fN[(long)q * r + t], f);
    rhs[(long)b * r + t] = fma(dt, f, acc);
  }
}

__global__ void sweep_store_kernel(const double* __restrict__ x, double* __restrict__ un, double* __restrict__ unm1,
                                   double* __restrict__ out, long step, long nt, int r, int keep_prev) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < r) {
    const double v = x[(long)b * r + t];
    if (keep_prev) unm1[(long)b * r + t] = un[(long)b * r + t];
    un[(long)b * r + t] = v;
    out[((long)b * nt + step) * r + t] = v;
  }
}

// ---- hyper-reduced sweep -----------------------------------------------------------------------------------
// G[b] = [ bdf F_mass | dt F_lin | dt S (W u* + C) ]  (one row of interpolation coefficients per parameter point)
// and G[B + b] = [ F_mass | 0 | 0 ], so that ONE skinny GEMM with Z yields K_N (rows 0..B-1) and M_N (rows B..2B-1)
// Optionally first closes a step (x -> u^n, u^n -> u^{n-1}, trajectory), then builds the coefficient rows of the
// next one from the updated state: one launch between two steps instead of two.
__global__ void hsweep_advance_kernel(const double* __restrict__ x, int do_store, int r, const rt_advance a) {
  extern __shared__ double su[];  // u* of the coming step
  hsweep_advance_rows(a, blockIdx.x, r, x, do_store, su, threadIdx.x, blockDim.x);
}

__global__ void hsweep_count_kernel(long* ctr) { *ctr += 1; }

// rhs[b] = M_N[b] (c0 u_n[b] + c1 u_nm1[b]) + dt Zf^T F_rhs[b]
__global__ void hsweep_rhs_kernel(const double* __restrict__ MN, const double* __restrict__ un,
                                  const double* __restrict__ unm1, double c0, double c1, double dt,
                                  const double* __restrict__ Ff, const double* __restrict__ Zf, int mf, int r,
                                  double* __restrict__ rhs) {
  extern __shared__ double su[];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < r) su[t] = c0 * un[(long)b * r + t] + c1 * unm1[(long)b * r + t];
  __syncthreads();
  if (t < r) {
    const double* M = MN + (long)b * r * r + (long)t * r;
    double acc = 0.0;
    for (int j = 0; j < r; ++j) acc = fma(M[j], su[j], acc);
    double f = 0.0;
    for (int e = 0; e < mf; ++e) f = fma(Ff[(long)b * mf + e], Zf[(long)e * r + t], f);
    rhs[(long)b * r + t] = fma(dt, f, acc);
  }
}

}  // namespace

extern "C" int rt_rom_bdf_sweep(rt_ctx* ctx, const rt_sweep_desc* d, double* uN_out) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, d && uN_out);
  RT_ARG_CHECK(ctx, d->N >= 1 && d->nnz >= 1 && d->r >= 1 && d->r <= 128 && d->n_mu >= 1 && d->nt >= 1);
  RT_ARG_CHECK(ctx, d->indptr && d->indices && d->V && d->mass_values && d->n_terms >= 0 && d->n_rhs >= 0);
  RT_ARG_CHECK(ctx, (d->n_terms == 0 || (d->term_values && d->term_coef)) && (d->n_rhs == 0 || (d->rhs_terms && d->rhs_coef)));
  const long N = d->N, nnz = d->nnz, r = d->r, B = d->n_mu, nt = d->nt;
  const int Q = (int)d->n_terms, F = (int)d->n_rhs;
  hipStream_t st = ctx->stream;

  // workspace (composite arena)
  size_t off = 0;
  auto take = [&off](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t oRow = take(sizeof(int) * nnz), oMN = take(sizeof(double) * r * r), oFN = take(sizeof(double) * (F ? F : 1) * r),
               oKN = take(sizeof(double) * B * r * r), oRhs = take(sizeof(double) * B * r), oUn = take(sizeof(double) * B * r),
               oUm = take(sizeof(double) * B * r), oUh = take(sizeof(double) * B * N), oUp = take(sizeof(double) * B * N),
               oXT = take(sizeof(double) * r * B), oInfo = take(sizeof(int) * B),
               oKv = take(sizeof(double) * B * nnz), oTab = take(rt_project_stage_table_bytes(N)),
               oXi = take(sizeof(double) * B * r * r);
  void* base = nullptr;
  int rc = rt_scratch2(ctx, off, &base);
  if (rc != RT_OK) return rc;
  char* b8 = static_cast<char*>(base);
  int* row_of = reinterpret_cast<int*>(b8 + oRow);
  double* MN = reinterpret_cast<double*>(b8 + oMN);
  double* fN = reinterpret_cast<double*>(b8 + oFN);
  double* KN = reinterpret_cast<double*>(b8 + oKN);
  double* rhs = reinterpret_cast<double*>(b8 + oRhs);
  double* un = reinterpret_cast<double*>(b8 + oUn);
  double* unm1 = reinterpret_cast<double*>(b8 + oUm);
  double* uh = reinterpret_cast<double*>(b8 + oUh);
  double* uhp = reinterpret_cast<double*>(b8 + oUp);
  double* xT = reinterpret_cast<double*>(b8 + oXT);
  int* info = reinterpret_cast<int*>(b8 + oInfo);
  double* kval = reinterpret_cast<double*>(b8 + oKv);
  double* Xinv = reinterpret_cast<double*>(b8 + oXi);  // K_N^-1 of the previous step, per parameter point
  void* stage_table = b8 + oTab;  // per-pattern stage records of the fused projection, built once per sweep

  RT_HIP_CHECK(ctx, hipMemsetAsync(un, 0, sizeof(double) * B * r, st));
  RT_HIP_CHECK(ctx, hipMemsetAsync(unm1, 0, sizeof(double) * B * r, st));
  RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->dev_counters + RT_CNT_NS_ITER, 0, sizeof(long) * 4, st));  // rt_last_sweep_stats
  RT_HIP_CHECK(ctx, hipMemsetAsync(uh, 0, sizeof(double) * B * N, st));
  RT_HIP_CHECK(ctx, hipMemsetAsync(uhp, 0, sizeof(double) * B * N, st));
  hipLaunchKernelGGL(sweep_rows_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<const long*>(d->indptr), N, row_of);
  RT_HIP_CHECK(ctx, hipGetLastError());
  rc = rt_project_stage_table(ctx, d->indptr, d->indices, N, stage_table);
  if (rc != RT_OK) return rc;
  int banded = -1;  // read the table's header back once: every step then launches only the kernel variant it needs
  rc = rt_project_stage_table_banded(ctx, stage_table, &banded);
  if (rc != RT_OK) return rc;
  // M_N = V^T M V and f_N[f] = V^T f_f, once
  rc = rt_project_fused(ctx, d->indptr, d->indices, d->mass_values, 1, 0, 1, N, d->V, r, r, MN, stage_table, banded);
  if (rc != RT_OK) return rc;
  if (F) {
    // rhs_terms is F x N (each vector contiguous): A(k, i) = V[k][i], B(k, f) = rhs_terms[f][k]
    rc = rt_gemm_strided(ctx, d->rhs_terms, 1, N, d->V, r, 1, N, F, r, fN, r, 1, false, true);
    if (rc != RT_OK) return rc;
  }

  for (long step = 0; step < nt; ++step) {
    const bool second = d->bdf2 && step > 0;
    if ((long)sizeof(double) * B * Q <= SV_LDS_BYTES)
      hipLaunchKernelGGL(sweep_values_kernel<true>, dim3((unsigned)((nnz + 255) / 256)), dim3(256), sizeof(double) * B * Q,
                         st, d->mass_values, d->term_values, Q, Q ? d->term_coef + step * B * Q : nullptr, d->tril_values,
                         row_of, uh, uhp, d->bdf2 ? 1 : 0, nnz, N, (int)B, second ? 1.5 : 1.0, d->dt, kval);
    else
      hipLaunchKernelGGL(sweep_values_kernel<false>, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st,
                         d->mass_values, d->term_values, Q, d->term_coef + step * B * Q, d->tril_values, row_of, uh, uhp,
                         d->bdf2 ? 1 : 0, nnz, N, (int)B, second ? 1.5 : 1.0, d->dt, kval);
    RT_HIP_CHECK(ctx, hipGetLastError());
    rc = rt_project_fused(ctx, d->indptr, d->indices, kval, 1, nnz, B, N, d->V, r, r, KN, stage_table, banded);
    if (rc != RT_OK) return rc;
    const double c0 = d->bdf2 ? 2.0 : 1.0, c1 = d->bdf2 ? -0.5 : 0.0;  // u^{n-1} = 0 at step 0 reproduces BDF1
    // reference: step 0 of a BDF2 run uses M_N (2 u^0 - u^{-1}/2) with both zero (rom.py:451-458,921-924)
    // consecutive K_N differ by O(dt): refresh the tracked inverse on the matrix cores; the solve kernel forms the
    // right-hand side, falls back to LU by itself where the tracking fails, and closes the step (u^n, u^{n-1},
    // trajectory, u^n transposed for the lift) - one launch where there were four
    rt_newton_rhs rq{MN, un, unm1, c0, c1, d->dt, F ? d->rhs_coef + step * B * F : nullptr, fN, F};
    rq.mn_stride = 0;
    rt_advance close{};
    close.un = un; close.unm1 = unm1; close.out = uN_out; close.step_done = step; close.nt = nt;
    close.keep_prev = d->bdf2 ? 1 : 0; close.do_coef = 0; close.xT = xT; close.B = (int)B; close.enabled = 1;
    rc = rt_newton_solve_batched(ctx, KN, Xinv, rhs, r, B, step > 0 ? 1 : 0, info, &rq, &close);
    if (rc == RT_ERR_UNSUPPORTED) {  // r > 80: the LDS-resident iteration does not fit
      hipLaunchKernelGGL(sweep_rhs_kernel, dim3((unsigned)B), dim3(128), sizeof(double) * r, st, MN, un, unm1, c0, c1,
                         d->dt, F ? d->rhs_coef + step * B * F : nullptr, fN, F, (int)r, rhs);
      RT_HIP_CHECK(ctx, hipGetLastError());
      rc = rt_dense_solve_batched(ctx, KN, rhs, r, B, info);
      if (rc != RT_OK) return rc;
      hipLaunchKernelGGL(sweep_store_kernel, dim3((unsigned)B), dim3(128), 0, st, rhs, un, unm1, uN_out, step, nt, (int)r,
                         d->bdf2 ? 1 : 0);
      RT_HIP_CHECK(ctx, hipGetLastError());
      rc = rt_transpose(ctx, un, B, r, r, xT, B);
    }
    if (rc != RT_OK) return rc;
    // u_h <- V u_N for every mu, stored [mu][N]; the previous u_h becomes u_h^{n-1}
    double* tmp = uhp; uhp = uh; uh = tmp;
    // Y (N x B, column-major ld N) = V (N x r) * xT (r x B); xT[j][b] = un[b][j]
    rc = rt_gemm_nn(ctx, d->V, r, RT_ROW_MAJOR, xT, B, N, r, B, uh, N, RT_COL_MAJOR);
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

// Hyper-reduced online sweep: every reduced operator is an interpolation expansion sum_e g_e Z_e whose
// coefficients g are the operator's own entries at its (M)DEIM entries (tables for the (mu,t)-dependent
// operators, an affine map of u_N* for the state-dependent one).  Per step: one small kernel builds the
// coefficient rows, two skinny GEMMs on the matrix cores give K_N and M_N for all parameter points, then the
// right-hand side, the reduced solve (inverse tracking) and the store.  Nothing of size N_h exists here.
extern "C" int rt_hrom_bdf_sweep(rt_ctx* ctx, const rt_hsweep_desc* d, double* uN_out) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, d && uN_out);
  RT_ARG_CHECK(ctx, d->r >= 1 && d->r <= 128 && d->n_mu >= 1 && d->nt >= 1);
  RT_ARG_CHECK(ctx, d->m_mass >= 1 && d->m_lin >= 0 && d->m_nl >= 0 && d->m_rhs >= 0 && d->Z && d->F_mass);
  RT_ARG_CHECK(ctx, (d->m_lin == 0 || d->F_lin) && (d->m_nl == 0 || d->W) && (d->m_rhs == 0 || (d->Zf && d->F_rhs)));
  const long r = d->r, B = d->n_mu, nt = d->nt, mm = d->m_mass, ml = d->m_lin, mn = d->m_nl, mf = d->m_rhs;
  const long M = mm + ml + mn, rr = r * r;
  hipStream_t st = ctx->stream;
  size_t off = 0;
  auto take = [&off](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t oG = take(sizeof(double) * 2 * B * M), oKN = take(sizeof(double) * 2 * B * rr),
               oRhs = take(sizeof(double) * B * r), oUn = take(sizeof(double) * B * r), oUm = take(sizeof(double) * B * r),
               oXi = take(sizeof(double) * B * rr), oInfo = take(sizeof(int) * B), oCtr = take(sizeof(long));
  void* base = nullptr;
  int rc = rt_scratch2(ctx, off, &base);
  if (rc != RT_OK) return rc;
  char* b8 = static_cast<char*>(base);
  double* G = reinterpret_cast<double*>(b8 + oG);
  double* KN = reinterpret_cast<double*>(b8 + oKN);
  double* MN = KN + B * rr;  // second half of the stacked product
  double* rhs = reinterpret_cast<double*>(b8 + oRhs);
  double* un = reinterpret_cast<double*>(b8 + oUn);
  double* unm1 = reinterpret_cast<double*>(b8 + oUm);
  double* Xinv = reinterpret_cast<double*>(b8 + oXi);
  int* info = reinterpret_cast<int*>(b8 + oInfo);
  long* ctr = reinterpret_cast<long*>(b8 + oCtr);
  RT_HIP_CHECK(ctx, hipMemsetAsync(un, 0, sizeof(double) * B * r, st));
  RT_HIP_CHECK(ctx, hipMemsetAsync(unm1, 0, sizeof(double) * B * r, st));
  RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->dev_counters + RT_CNT_NS_ITER, 0, sizeof(long) * 4, st));  // rt_last_sweep_stats

  // the end-of-step work for closing step next-1 (if any) and preparing the rows of step `next`
  auto advance_args = [&](long next) {
    const bool has_next = next < nt;
    const long s2 = has_next ? next : 0;
    rt_advance a{};
    a.un = un; a.unm1 = unm1; a.out = uN_out; a.step_done = next - 1; a.nt = nt;
    a.keep_prev = d->bdf2 ? 1 : 0; a.do_coef = has_next ? 1 : 0;
    a.Fm = d->F_mass + s2 * B * mm; a.Fl = ml ? d->F_lin + s2 * B * ml : nullptr; a.W = d->W;
    a.Cn = d->C_nl ? d->C_nl + s2 * B * mn : nullptr; a.Sn = d->S_nl ? d->S_nl + s2 * B : nullptr;
    a.extrapolate = d->bdf2 ? 1 : 0; a.mm = (int)mm; a.ml = (int)ml; a.mn = (int)mn;
    a.bdf = (d->bdf2 && next > 0) ? 1.5 : 1.0; a.dt = d->dt; a.G = G; a.ctr = nullptr; a.B = (int)B; a.enabled = 1;
    return a;
  };
  auto advance = [&](long next, int do_store) {
    hipLaunchKernelGGL(hsweep_advance_kernel, dim3((unsigned)B), dim3(256), sizeof(double) * r, st, rhs, do_store, (int)r,
                       advance_args(next));
  };
  advance(0, 0);
  RT_HIP_CHECK(ctx, hipGetLastError());
  const double c0 = d->bdf2 ? 2.0 : 1.0, c1 = d->bdf2 ? -0.5 : 0.0;
  // Steps 1 .. nt-1 are the same two launches (expansion GEMM, tracked solve with the end of the step in its tail) with
  // only table offsets moving.  With rt_ctx_set_option(ctx, "sweep_graph", 1) they are captured ONCE as a hipGraph whose
  // kernels take the step from a device counter, and replayed; step 0 runs eagerly (first BDF step, inverse tracking
  // starts, scratch arenas get their sizes).  It is an option, not the default: on the pool's boxes plain launches
  // are faster (60 us per step against 69-86 us replayed: the graph's own inter-node gaps and the counter kernel).
  // Not with r > 80 (no inverse tracking) or in profile mode (event pairs inside the GEMM).
  const bool use_graph = ctx->sweep_graph && nt > 2 && r <= 80 && !ctx->profile;
  for (long step = 0; step < nt; ++step) {
    if (use_graph && step == 1) {
      const long one = 1;
      RT_HIP_CHECK(ctx, hipMemcpyAsync(ctr, &one, sizeof(long), hipMemcpyHostToDevice, st));
      RT_HIP_CHECK(ctx, hipStreamSynchronize(st));  // `one` is on this stack frame
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      // the caller's stream may be the legacy default stream, which cannot capture: the graph is captured and
      // replayed on a stream of its own (everything before it has completed - the synchronisation above - and the
      // sweep returns only when the replay has)
      hipStream_t gs = nullptr;
      RT_HIP_CHECK(ctx, hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
      struct Restore {
        rt_ctx* c; hipStream_t keep, mine;
        ~Restore() { c->stream = keep; (void)hipStreamDestroy(mine); }
      } restore{ctx, st, gs};
      ctx->stream = gs;
      st = gs;
      RT_HIP_CHECK(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      rc = rt_expansion_gemm(ctx, G, M, d->Z, rr, KN, rr, 2 * B, M, rr);
      if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm_strided(ctx, G, 1, M, d->Z, rr, 1, M, 2 * B, rr, KN, rr, 1, false, false);
      rt_newton_rhs rq{MN, un, unm1, c0, c1, d->dt, mf ? d->F_rhs : nullptr, d->Zf, (int)mf};
      rq.ctr = ctr;
      rq.ff_stride = B * mf;
      rt_advance ga = advance_args(1);   // table bases + device counter: the replayed launches cannot carry the step
      ga.Fm = d->F_mass; ga.Fl = ml ? d->F_lin : nullptr; ga.Cn = d->C_nl; ga.Sn = d->S_nl;
      ga.bdf = d->bdf2 ? 1.5 : 1.0; ga.do_coef = 1; ga.ctr = ctr;
      if (rc == RT_OK) rc = rt_newton_solve_batched(ctx, KN, Xinv, rhs, r, B, 1, info, &rq, &ga);
      hipLaunchKernelGGL(hsweep_count_kernel, dim3(1), dim3(1), 0, st, ctr);
      const hipError_t cap = hipStreamEndCapture(st, &graph);
      if (rc != RT_OK || cap != hipSuccess || graph == nullptr) {
        if (graph) (void)hipGraphDestroy(graph);
        ctx->err = "rt_hrom_bdf_sweep: capturing the step graph failed";
        return rc != RT_OK ? rc : RT_ERR_HIP;
      }
      {
        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) {
          (void)hipGraphDestroy(graph);
          RT_HIP_CHECK(ctx, ie);
        }
      }
      hipError_t le = hipSuccess;
      for (long s2 = 1; s2 < nt && le == hipSuccess; ++s2) le = hipGraphLaunch(exec, st);
      const hipError_t se = hipStreamSynchronize(st);  // the executable graph owns the launch arguments: keep it until done
      if (le == hipSuccess) le = se;
      (void)hipGraphExecDestroy(exec);
      (void)hipGraphDestroy(graph);
      RT_HIP_CHECK(ctx, le);
      break;
    }
    // [K_N; M_N][b][ij] = sum_e G[b][e] Z[e][ij]  (2 B rows)
    rc = rt_expansion_gemm(ctx, G, M, d->Z, rr, KN, rr, 2 * B, M, rr);
    if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm_strided(ctx, G, 1, M, d->Z, rr, 1, M, 2 * B, rr, KN, rr, 1, false, false);
    if (rc != RT_OK) return rc;
    rt_newton_rhs rq{MN, un, unm1, c0, c1, d->dt, mf ? d->F_rhs + step * B * mf : nullptr, d->Zf, (int)mf};
    // two launches per step: the expansion GEMM and the solve, which forms the right-hand side, falls back to a
    // pivoted LU by itself where the tracked inverse fails, and closes the step (state, trajectory, next rows of G)
    const rt_advance adv = advance_args(step + 1);
    rc = rt_newton_solve_batched(ctx, KN, Xinv, rhs, r, B, step > 0 ? 1 : 0, info, &rq, &adv);
    if (rc == RT_OK) {
      RT_HIP_CHECK(ctx, hipGetLastError());
      continue;
    } else if (rc == RT_ERR_UNSUPPORTED) {  // r > 80: right-hand side by its own kernel, then the LU
      hipLaunchKernelGGL(hsweep_rhs_kernel, dim3((unsigned)B), dim3(128), sizeof(double) * r, st, MN, un, unm1, c0, c1,
                         d->dt, mf ? d->F_rhs + step * B * mf : nullptr, d->Zf, (int)mf, (int)r, rhs);
      rc = rt_dense_solve_batched(ctx, KN, rhs, r, B, info);
    }
    if (rc != RT_OK) return rc;
    advance(step + 1, 1);
    RT_HIP_CHECK(ctx, hipGetLastError());
  }
  return RT_OK;
}
