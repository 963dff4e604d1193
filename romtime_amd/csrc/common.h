// Shared declarations for libromtime_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/romtime_hip.h"

constexpr int RT_N_COUNTERS = 8;

struct rt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  void* scratch = nullptr;   // leaf arena: GEMM split slabs
  size_t scratch_bytes = 0;
  void* scratch2 = nullptr;  // composite arena: workspaces of ops that call the GEMM (AV, DEIM state)
  size_t scratch2_bytes = 0;
  int num_cus = 256;      // CUs this ctx sizes its grids for: the device's, or fewer when its stream is CU-masked ("cu_limit")
  int device_cus = 256;
  std::string err;
  int64_t last_grid = 0, last_splits = 0, last_tile = 0;
  bool profile = false;             // bracket the main GEMM kernel with events (rt_ctx_set_profile)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  hipEvent_t gev0 = nullptr, gev1 = nullptr;  // the Gram kernel's own pair (rt_last_gram_ms): survives later GEMMs
  bool gev_valid = false;
  bool sweep_graph = false;         // rt_hrom_bdf_sweep replays steps 1.. as a hipGraph (rt_ctx_set_option)
  bool eig_one_xcd = true;          // allow the one-XCD form of the eigensolver's hand-off (rt_ctx_set_option)
  int eig_xcd = 0;                  // ... and the XCD its workers are put on ("eig_xcd": eigensolves on different XCDs run side by side)
  // state handed from rt_sym_eig_values to rt_sym_eig_vectors (pointers into the composite arena)
  struct {
    const double *d = nullptr, *e = nullptr, *V = nullptr, *tau = nullptr;
    int64_t n = 0;
    void* base = nullptr;
    uint64_t gen = 0;  // scratch2_gen when the reflectors were written: any later user of the arena invalidates them
  } eig;
  uint64_t scratch2_gen = 0;         // bumped by every rt_scratch2 call (each one hands the arena to a new owner)
  std::vector<const void*> lds_done; // kernels whose dynamic-LDS limit this ctx has raised (rt_func_lds)
  // Device-side event counters (hipMalloc'ed with the ctx, bumped by the kernels themselves because the host never
  // waits for them): rt_ctx_get_counter / rt_last_sweep_stats copy them back.
  long* dev_counters = nullptr;
  // progress counters of the snapshot Gram kernels (gram_mfma.hip: per launch kind and XCD, {stages done, workgroups
  // started}, 128 B apart); zeroed when allocated and again by every Gram's reduction kernel
  unsigned long long* gram_pace = nullptr;
  bool gram_pace_on = true;         // "gram_pace" option
};

// slots of rt_ctx::dev_counters
enum {
  RT_CNT_EIG_TIMEOUT = 0,       // hand-offs of the tridiagonalisation that hit the wall-clock bound
  RT_CNT_EIG_GENERAL_FORM = 1,  // tridiagonalisations that ran the write-through hand-off
  RT_CNT_EIG_ONE_XCD = 2,       // ... the one-XCD hand-off
  RT_CNT_GRAM_OFF_XCD = 3,      // workgroups of the snapshot Gram kernel that ran on another XCD than blockIdx % 8
  RT_CNT_NS_ITER = 4,           // online sweep: Newton-Schulz iterations (zeroed at the start of every sweep)
  RT_CNT_NS_RESTART = 5,        // ... systems restarted from K^T / (|K|_1 |K|_inf)
  RT_CNT_LU_FALLBACK = 6,       // ... systems handed to the pivoted LU
  RT_CNT_SOLVES = 7             // ... systems solved
};

#define RT_TRY(expr)                 \
  do {                               \
    const int _rc = (expr);          \
    if (_rc != RT_OK) return _rc;    \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the function object of the CURRENT device.  A ctx is bound
// to one device (and one host thread), so it raises the limit once per kernel and remembers it; process-wide
// "done" flags would leave the second GPU of a single-process multi-GPU host at the default 64 KB.
int rt_func_lds(rt_ctx* ctx, const void* fn, int bytes);

#define RT_HIP_CHECK(ctx, expr)                                                              \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                        \
      return RT_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

#define RT_ARG_CHECK(ctx, cond)                                                              \
  do {                                                                                       \
    if (!(cond)) {                                                                           \
      (ctx)->err = std::string("bad argument: ") + #cond;                                    \
      return RT_ERR_ARG;                                                                     \
    }                                                                                        \
  } while (0)

// Returns a device pointer to at least `bytes` of scratch owned by the ctx (256-B aligned).
int rt_scratch(rt_ctx* ctx, size_t bytes, void** out);
int rt_scratch2(rt_ctx* ctx, size_t bytes, void** out);

// Generic strided GEMM on the f64 matrix cores:  C(i,j) = sum_k A(k,i) * B(k,j).
//   A(k,i) at A[k*a_ks + i*a_ms]  (exactly one of a_ks / a_ms is 1, or M == 1)
//   B(k,j) at B[k*b_ks + j*b_ns]
//   C(i,j) at C[i*c_rs + j*c_cs]
// `symmetric`: A == B, M == Nn; only tiles on/above the diagonal are computed, then mirrored.
// `allow_split`: contraction may be split over workgroups (deterministic slab reduction).
// `alpha`, `beta`: C = alpha * (product) + beta * C (beta == 0: C is not read).
// C (R x N) = G (R x K) Z (K x N), all row-major, R <= 64, short contraction (sweep.hip); RT_ERR_UNSUPPORTED otherwise
int rt_expansion_gemm(rt_ctx* ctx, const double* G, long ldg, const double* Z, long ldz, double* Cm, long ldc, long R, long K,
                      long N);
int rt_gemm_strided(rt_ctx* ctx, const double* A, int64_t a_ks, int64_t a_ms, const double* B, int64_t b_ks,
                    int64_t b_ns, int64_t K, int64_t M, int64_t Nn, double* C, int64_t c_rs, int64_t c_cs,
                    bool symmetric, bool allow_split, double alpha = 1.0, double beta = 0.0);

// Specialised Gram kernel (gram_mfma.hip); RT_ERR_UNSUPPORTED means "use rt_gemm_strided".
int rt_gram128(rt_ctx* ctx, const double* X, int64_t ks, int64_t ms, int64_t K, int64_t n, double* G);

// Tall-skinny Y = X T for row-major X, k <= 64 (tallskinny.hip); RT_ERR_UNSUPPORTED outside its range.
int rt_tallskinny(rt_ctx* ctx, const double* X, int64_t ldx, const double* T, int64_t ldt, int64_t N, int64_t n,
                  int64_t k, double* Y, int64_t ldy);

// C = A^T B for few columns of A (rank_update.hip); RT_ERR_UNSUPPORTED outside its range.
int rt_skinny_tn(rt_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, int64_t N, int64_t m, int64_t n,
                 double* Cm, int64_t ldc);

// Newton-Schulz inverse tracking solve for the online sweep (solve.hip); RT_ERR_UNSUPPORTED for r > 80.
struct rt_newton_rhs {  // b = M_N (c0 u^n + c1 u^{n-1}) + dt Zf^T F_rhs, per system; MN == nullptr: rhs is given
  const double* MN;    // B x r x r
  const double* un;    // B x r
  const double* unm1;  // B x r
  double c0, c1, dt;
  const double* Ff;    // B x mf
  const double* Zf;    // mf x r
  int mf;
  long mn_stride = -1;        // doubles between the M_N of consecutive systems (-1: r * r; 0: one M_N for all)
  const long* ctr = nullptr;  // device step counter: Ff is the table base and the step's rows start at *ctr * ff_stride
  long ff_stride = 0;         // (graph replay of a sweep: the launch parameters cannot carry the step)
};
struct rt_advance;  // sweep_advance.h: the hyper-reduced sweep's end-of-step work, run as the tail of the solver kernels
int rt_newton_solve_batched(rt_ctx* ctx, const double* K, double* Xinv, double* rhs, int64_t r, int64_t B,
                            int have_prev, int* info, const rt_newton_rhs* recipe = nullptr,
                            const rt_advance* advance = nullptr);

// Fused SpMM + V^T(.) projection (project_fused.hip); RT_ERR_UNSUPPORTED for r > 128.  `stage_table` is the
// per-pattern table built by rt_project_stage_table (rt_project_stage_table_bytes(N) bytes of device memory), or
// nullptr to have it built on every call.
size_t rt_project_stage_table_bytes(int64_t N);
int rt_project_stage_table(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, int64_t N, void* table);
// `banded`: 1 / 0 when the caller has read the table's header (rt_project_stage_table_banded) and only the kernel
// variant for that kind of pattern need be launched; -1: both are launched and the wrong one returns at once.
int rt_project_fused(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, const double* data, int64_t d_es,
                     int64_t d_bs, int64_t B, int64_t N, const double* V, int64_t ldv, int64_t r, double* AN,
                     const void* stage_table = nullptr, int banded = -1);
int rt_project_stage_table_banded(rt_ctx* ctx, const void* table, int* banded);   // synchronises the ctx stream
