// Context management and the thin C-ABI wrappers around the kernels.
#include "common.h"

#include <chrono>
#include <new>

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

int grow(rt_ctx* ctx, void** buf, size_t* have, size_t bytes, void** out) {
  if (bytes > *have) {
    // kernels already enqueued may still use the old block
    RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (*buf) RT_HIP_CHECK(ctx, hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    RT_HIP_CHECK(ctx, hipMalloc(buf, want));
    *have = want;
  }
  *out = *buf;
  return RT_OK;
}

__global__ void gram_scale_kernel(double* __restrict__ G, int n, double* __restrict__ colnorm, int normalize,
                                  int* status_flag) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const int i = (int)(idx / n), j = (int)(idx % n);
  const double di = sqrt(G[(long)i * n + i]), dj = sqrt(G[(long)j * n + j]);
  if (i == j) {
    colnorm[i] = di;
    if (status_flag && !(di > 0.0)) *status_flag = RT_WARN_ZERO_NORM;
  }
  // the diagonal is only read here; it is normalised by a second launch (gram_unit_diag_kernel)
  if (normalize && i != j) G[idx] = G[idx] / (di * dj);
}

__global__ void gram_unit_diag_kernel(double* __restrict__ G, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double d = G[(long)i * n + i];
    G[(long)i * n + i] = d / (sqrt(d) * sqrt(d));
  }
}

__global__ void transpose_kernel(const double* __restrict__ src, long rows, long cols, long ld_src,
                                 double* __restrict__ dst, long ld_dst) {
  __shared__ double tile[32][33];
  const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 32;
  for (int rr = threadIdx.y; rr < 32; rr += blockDim.y) {
    const long r = r0 + rr, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[rr][threadIdx.x] = src[r * ld_src + c];
  }
  __syncthreads();
  for (int cc = threadIdx.y; cc < 32; cc += blockDim.y) {
    const long c = c0 + cc, r = r0 + threadIdx.x;
    if (c < cols && r < rows) dst[c * ld_dst + r] = tile[threadIdx.x][cc];
  }
}

// NACC independent accumulator chains per wave; operands vary per lane so the data is not trivial
template <int NACC>
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(int iters, double* sink) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double x = 1.0 + threadIdx.x * 1.1e-3, y = 0.7 - threadIdx.x * 0.9e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64((i & 1) ? x : y, (i & 2) ? x : y, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == -1.2345) sink[0] = s;
}

// Co-issue probe: 8 waves, waves w and w + 4 share a SIMD.  mode bit 0: waves 0-3 run `iters` x 4 independent
// FP64 MFMAs; bit 1: waves 4-7 run `iters` x 16 v_fma_f64; bit 2: waves 4-7 run `iters` x 16 integer VALU ops;
// bit 3 (mode 8+): waves 4-7 run `iters` x 16 LDS reads.  Elapsed times of the modes tell whether VALU / LDS
// work of one wave overlaps with the MFMAs of another wave on the same SIMD.
__global__ __launch_bounds__(512) void coissue_kernel(int iters, int mode, double* sink) {
  __shared__ double lds[2048];
  const int wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = 1.0 + i * 1e-6;
  __syncthreads();
  double s = 0.0;
  if (wid < 4) {
    if (mode & 1) {
      d4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
      double x = 1.0 + threadIdx.x * 1.1e-3, y = 0.7 - threadIdx.x * 0.9e-3;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64((i & 1) ? x : y, (i & 2) ? x : y, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    if (mode & 2) {
      double a[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = 1.0 + threadIdx.x * 1e-3 * i;
      const double m = 1.0000001, c = 1e-9;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] = fma(a[i], m, c);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += a[i];
    }
    if (mode & 4) {
      unsigned a[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] = (a[i] ^ (a[i] >> 3)) + 0x9e3779b9u;  // 2 integer VALU ops... counted as 2
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += (double)a[i];
    }
    if (mode & 8) {
      int idx = threadIdx.x & 63;
      double a = 0.0;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a += lds[(idx + 64 * i) & 2047];
        idx = (idx + 1) & 63;
      }
      s += a;
    }
  }
  if (s == -1.2345) sink[0] = s;
}

__global__ __launch_bounds__(256) void copy_kernel(double4* __restrict__ dst, const double4* __restrict__ src, long n4) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace

int rt_scratch(rt_ctx* ctx, size_t bytes, void** out) { return grow(ctx, &ctx->scratch, &ctx->scratch_bytes, bytes, out); }
int rt_scratch2(rt_ctx* ctx, size_t bytes, void** out) {
  ++ctx->scratch2_gen;  // the arena changes hands: state an earlier call left in it (the eigensolver's reflectors) is void
  return grow(ctx, &ctx->scratch2, &ctx->scratch2_bytes, bytes, out);
}

int rt_func_lds(rt_ctx* ctx, const void* fn, int bytes) {
  for (const void* f : ctx->lds_done)
    if (f == fn) return RT_OK;
  RT_HIP_CHECK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  ctx->lds_done.push_back(fn);
  return RT_OK;
}

extern "C" {

int rt_version(void) { return 320; }  // 320: rt_gram_plan_info; 310: option gram_pace (paced / one-launch Gram); 300 (round 3): rt_dense_solve_multi, RT_P1_LOAD_P2; 210: rt_tracked_solve_batched, rt_pod_enqueue, options eig_xcd, counter gram_off_xcd

int rt_ctx_create(rt_ctx** out, int device) {
  if (!out) return RT_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return RT_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return RT_ERR_HIP;
  rt_ctx* ctx = new (std::nothrow) rt_ctx();
  if (!ctx) return RT_ERR_HIP;
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = ctx->device_cus = prop.multiProcessorCount;
  if (hipMalloc(reinterpret_cast<void**>(&ctx->dev_counters), sizeof(long) * RT_N_COUNTERS) != hipSuccess ||
      hipMemset(ctx->dev_counters, 0, sizeof(long) * RT_N_COUNTERS) != hipSuccess) {
    delete ctx;
    return RT_ERR_HIP;
  }
  *out = ctx;
  return RT_OK;
}

void rt_ctx_destroy(rt_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  // the whole device, not ctx->stream: the stream last bound to this ctx may be gone already (a CU-masked stream
  // destroyed by the pipeline's shutdown before the interpreter drops the ctx), and hipFree below waits for the device anyway
  (void)hipDeviceSynchronize();
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->scratch2) (void)hipFree(ctx->scratch2);
  if (ctx->dev_counters) (void)hipFree(ctx->dev_counters);
  if (ctx->gram_pace) (void)hipFree(ctx->gram_pace);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->gev0) (void)hipEventDestroy(ctx->gev0);
  if (ctx->gev1) (void)hipEventDestroy(ctx->gev1);
  delete ctx;
}

int rt_ctx_set_stream(rt_ctx* ctx, void* hip_stream) {
  if (!ctx) return RT_ERR_ARG;
  ctx->stream = static_cast<hipStream_t>(hip_stream);
  return RT_OK;
}

// CU-partitioned streams.  On MI355X bit i of a queue's CU mask is CU i / 8 of XCD i % 8, and a mask that leaves an XCD
// without CUs is ignored (measured: tools/probes/probe_cumask.hip), so a partition gives every XCD's CUs
// [first, first + count) to one stream and the rest to another: kernels of the two then run side by side on disjoint
// CUs whatever their launch order (no co-residency assumptions, no spinning on placement).
int rt_stream_create_cu_range(int device, int first_cu_per_xcd, int n_cu_per_xcd, void** stream) {
  if (!stream) return RT_ERR_ARG;
  *stream = nullptr;
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return RT_ERR_HIP;
  const int per_xcd = prop.multiProcessorCount / 8;
  if (first_cu_per_xcd < 0 || n_cu_per_xcd < 1 || first_cu_per_xcd + n_cu_per_xcd > per_xcd) return RT_ERR_ARG;
  const int words = (prop.multiProcessorCount + 31) / 32;
  std::vector<uint32_t> mask((size_t)words, 0u);
  for (int cu = first_cu_per_xcd; cu < first_cu_per_xcd + n_cu_per_xcd; ++cu)
    for (int xcd = 0; xcd < 8; ++xcd) {
      const int bit = cu * 8 + xcd;
      mask[bit >> 5] |= 1u << (bit & 31);
    }
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()) != hipSuccess) return RT_ERR_HIP;
  *stream = st;
  return RT_OK;
}

int rt_stream_destroy(void* stream) {
  if (!stream) return RT_ERR_ARG;
  return hipStreamDestroy(static_cast<hipStream_t>(stream)) == hipSuccess ? RT_OK : RT_ERR_HIP;
}

int rt_ctx_synchronize(rt_ctx* ctx) {
  if (!ctx) return RT_ERR_ARG;
  RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

const char* rt_last_error(rt_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int rt_last_launch_info(rt_ctx* ctx, int64_t* info3) {
  if (!ctx || !info3) return RT_ERR_ARG;
  info3[0] = ctx->last_grid;
  info3[1] = ctx->last_splits;
  info3[2] = ctx->last_tile;
  return RT_OK;
}

int rt_ctx_set_option(rt_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return RT_ERR_ARG;
  const std::string key(name);
  if (key == "eig_one_xcd") {
    ctx->eig_one_xcd = value != 0;
    return RT_OK;
  }
  if (key == "eig_xcd") {
    if (value < 0 || value > 7) {
      ctx->err = "rt_ctx_set_option: eig_xcd must be 0 .. 7";
      return RT_ERR_ARG;
    }
    ctx->eig_xcd = value;
    return RT_OK;
  }
  if (key == "gram_pace") {
    ctx->gram_pace_on = value != 0;
    return RT_OK;
  }
  if (key == "sweep_graph") {
    ctx->sweep_graph = value != 0;
    return RT_OK;
  }
  if (key == "cu_limit") {  // 0 = the whole device
    if (value < 0 || value > ctx->device_cus || (value % 8) != 0) {
      ctx->err = "rt_ctx_set_option: cu_limit must be a multiple of 8 between 0 and the device's CU count";
      return RT_ERR_ARG;
    }
    ctx->num_cus = value ? value : ctx->device_cus;
    return RT_OK;
  }
  ctx->err = "rt_ctx_set_option: unknown option " + key;
  return RT_ERR_ARG;
}

int rt_ctx_get_counter(rt_ctx* ctx, const char* name, int64_t* value) {
  if (!ctx || !name || !value) return RT_ERR_ARG;
  const std::string key(name);
  static const struct { const char* name; int slot; } table[] = {
      {"eig_timeouts", RT_CNT_EIG_TIMEOUT}, {"eig_general_form", RT_CNT_EIG_GENERAL_FORM},
      {"eig_one_xcd", RT_CNT_EIG_ONE_XCD}, {"gram_off_xcd", RT_CNT_GRAM_OFF_XCD}, {"sweep_newton_iterations", RT_CNT_NS_ITER},
      {"sweep_restarts", RT_CNT_NS_RESTART}, {"sweep_lu_fallbacks", RT_CNT_LU_FALLBACK}, {"sweep_solves", RT_CNT_SOLVES}};
  for (const auto& t : table)
    if (key == t.name) {
      long host = 0;
      RT_HIP_CHECK(ctx, hipMemcpyAsync(&host, ctx->dev_counters + t.slot, sizeof(long), hipMemcpyDeviceToHost, ctx->stream));
      RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      *value = host;
      return RT_OK;
    }
  ctx->err = "rt_ctx_get_counter: unknown counter " + key;
  return RT_ERR_ARG;
}

int rt_last_sweep_stats(rt_ctx* ctx, int64_t* stats4) {
  if (!ctx || !stats4) return RT_ERR_ARG;
  long host[4];
  RT_HIP_CHECK(ctx, hipMemcpyAsync(host, ctx->dev_counters + RT_CNT_NS_ITER, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 4; ++i) stats4[i] = host[i];
  return RT_OK;
}

int rt_ctx_set_profile(rt_ctx* ctx, int on) {
  if (!ctx) return RT_ERR_ARG;
  ctx->profile = (on != 0);
  ctx->ev_valid = false;
  ctx->gev_valid = false;
  return RT_OK;
}

int rt_last_gram_ms(rt_ctx* ctx, double* ms) {
  if (!ctx || !ms) return RT_ERR_ARG;
  if (!ctx->gev_valid) {
    ctx->err = "rt_last_gram_ms: no profiled launch of the snapshot Gram kernel recorded";
    return RT_ERR_ARG;
  }
  RT_HIP_CHECK(ctx, hipEventSynchronize(ctx->gev1));
  float f = 0.f;
  RT_HIP_CHECK(ctx, hipEventElapsedTime(&f, ctx->gev0, ctx->gev1));
  *ms = f;
  return RT_OK;
}

int rt_last_gemm_ms(rt_ctx* ctx, double* ms) {
  if (!ctx || !ms) return RT_ERR_ARG;
  if (!ctx->ev_valid) {
    ctx->err = "rt_last_gemm_ms: no profiled GEMM launch recorded";
    return RT_ERR_ARG;
  }
  RT_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
  float f = 0.f;
  RT_HIP_CHECK(ctx, hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
  *ms = f;
  return RT_OK;
}

int rt_gram(rt_ctx* ctx, const double* X, int64_t n_rows, int64_t n_cols, int64_t ld, int layout, double* G) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, X && G && n_rows >= 1 && n_cols >= 1);
  RT_ARG_CHECK(ctx, layout == RT_ROW_MAJOR || layout == RT_COL_MAJOR);
  RT_ARG_CHECK(ctx, ld >= (layout == RT_ROW_MAJOR ? n_cols : n_rows));
  const int64_t ks = (layout == RT_ROW_MAJOR) ? ld : 1, ms = (layout == RT_ROW_MAJOR) ? 1 : ld;
  const int rc = rt_gram128(ctx, X, ks, ms, n_rows, n_cols, G);
  if (rc != RT_ERR_UNSUPPORTED) return rc;
  return rt_gemm_strided(ctx, X, ks, ms, X, ks, ms, n_rows, n_cols, n_cols, G, n_cols, 1, true, true);
}

int rt_gram_scale(rt_ctx* ctx, double* G, int64_t n, double* colnorm, int normalize, int* status_flag) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, G && colnorm && n >= 1 && n <= 46340);
  if (status_flag) RT_HIP_CHECK(ctx, hipMemsetAsync(status_flag, 0, sizeof(int), ctx->stream));
  const long total = (long)n * n;
  hipLaunchKernelGGL(gram_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, G, (int)n,
                     colnorm, normalize, status_flag);
  if (normalize)
    hipLaunchKernelGGL(gram_unit_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, G, (int)n);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

int rt_gemm_tn(rt_ctx* ctx, const double* A, int64_t lda, int a_layout, const double* B, int64_t ldb, int b_layout,
               int64_t N, int64_t m, int64_t n, double* C, int64_t ldc) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, A && B && C && N >= 1 && m >= 1 && n >= 1 && ldc >= n);
  RT_ARG_CHECK(ctx, (a_layout == RT_ROW_MAJOR || a_layout == RT_COL_MAJOR) &&
                        (b_layout == RT_ROW_MAJOR || b_layout == RT_COL_MAJOR));
  RT_ARG_CHECK(ctx, lda >= (a_layout == RT_ROW_MAJOR ? m : N) && ldb >= (b_layout == RT_ROW_MAJOR ? n : N));
  const int64_t a_ks = (a_layout == RT_ROW_MAJOR) ? lda : 1, a_ms = (a_layout == RT_ROW_MAJOR) ? 1 : lda;
  const int64_t b_ks = (b_layout == RT_ROW_MAJOR) ? ldb : 1, b_ns = (b_layout == RT_ROW_MAJOR) ? 1 : ldb;
  const bool sym = (A == B) && (m == n) && (lda == ldb) && (a_layout == b_layout);
  if (!sym && a_layout == RT_ROW_MAJOR && b_layout == RT_ROW_MAJOR) {
    const int rc = rt_skinny_tn(ctx, A, lda, B, ldb, N, m, n, C, ldc);  // few modes against a tall copy: streaming kernel
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  return rt_gemm_strided(ctx, A, a_ks, a_ms, B, b_ks, b_ns, N, m, n, C, ldc, 1, sym, true);
}

int rt_gemm_nn(rt_ctx* ctx, const double* X, int64_t ldx, int x_layout, const double* T, int64_t ldt, int64_t N,
               int64_t n, int64_t k, double* Y, int64_t ldy, int y_layout) {
  return rt_gemm_nn_axpby(ctx, X, ldx, x_layout, T, ldt, N, n, k, 1.0, 0.0, Y, ldy, y_layout);
}

int rt_gemm_nn_axpby(rt_ctx* ctx, const double* X, int64_t ldx, int x_layout, const double* T, int64_t ldt, int64_t N,
                     int64_t n, int64_t k, double alpha, double beta, double* Y, int64_t ldy, int y_layout) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, X && T && Y && N >= 1 && n >= 1 && k >= 1 && ldt >= k);
  RT_ARG_CHECK(ctx, (x_layout == RT_ROW_MAJOR || x_layout == RT_COL_MAJOR) &&
                        (y_layout == RT_ROW_MAJOR || y_layout == RT_COL_MAJOR));
  RT_ARG_CHECK(ctx, ldx >= (x_layout == RT_ROW_MAJOR ? n : N) && ldy >= (y_layout == RT_ROW_MAJOR ? k : N));
  if (x_layout == RT_ROW_MAJOR && y_layout == RT_ROW_MAJOR && alpha == 1.0 && beta == 0.0) {
    const int rc = rt_tallskinny(ctx, X, ldx, T, ldt, N, n, k, Y, ldy);  // tall X, few columns out: streaming kernel
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  if (x_layout == RT_ROW_MAJOR && y_layout == RT_ROW_MAJOR && alpha == 1.0 && beta == 0.0 && N <= 64 && k >= 256 && n <= 2048) {
    const int rc = rt_expansion_gemm(ctx, X, ldx, T, ldt, Y, ldy, N, n, k);  // few rows, short contraction, wide output
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  // contraction over the n columns of X: A(c, i) = X[i][c]
  const int64_t a_ks = (x_layout == RT_ROW_MAJOR) ? 1 : ldx, a_ms = (x_layout == RT_ROW_MAJOR) ? ldx : 1;
  const int64_t c_rs = (y_layout == RT_ROW_MAJOR) ? ldy : 1, c_cs = (y_layout == RT_ROW_MAJOR) ? 1 : ldy;
  return rt_gemm_strided(ctx, X, a_ks, a_ms, T, ldt, 1, n, N, k, Y, c_rs, c_cs, false, false, alpha, beta);
}

int rt_transpose(rt_ctx* ctx, const double* src, int64_t rows, int64_t cols, int64_t ld_src, double* dst,
                 int64_t ld_dst) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, src && dst && rows >= 1 && cols >= 1 && ld_src >= cols && ld_dst >= rows);
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
  RT_ARG_CHECK(ctx, grid.y <= 65535 * 32);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, ctx->stream, src, (long)rows, (long)cols, (long)ld_src,
                     dst, (long)ld_dst);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

int rt_bench_mfma_f64(rt_ctx* ctx, int iters, double* tflops) {
  // iters encodes the variant: low 24 bits = iterations, bits 24..27 = log2(accumulators) (0 -> 4),
  // bits 28..30 = workgroups per CU (0 -> 2)
  if (!ctx || !tflops || iters < 1) return RT_ERR_ARG;
  const int n_it = iters & 0xffffff;
  const int lacc = (iters >> 24) & 15, wgcu = ((iters >> 28) & 7) ? ((iters >> 28) & 7) : 2;
  const int nacc = lacc ? (1 << lacc) : 4;
  void* sink = nullptr;
  int rc = rt_scratch(ctx, 256, &sink);
  if (rc != RT_OK) return rc;
  const int grid = ctx->num_cus * wgcu;
  hipEvent_t e0, e1;
  RT_HIP_CHECK(ctx, hipEventCreate(&e0));
  RT_HIP_CHECK(ctx, hipEventCreate(&e1));
  const bool coissue = (lacc == 15);  // diagnostic: *tflops receives the elapsed milliseconds of coissue_kernel
  auto launch = [&](int its) {
    double* sk = static_cast<double*>(sink);
    if (coissue) {
      hipLaunchKernelGGL(coissue_kernel, dim3(ctx->num_cus), dim3(512), 0, ctx->stream, its, (iters >> 28) & 7, sk);
      return;
    }
    switch (nacc) {
      case 2: hipLaunchKernelGGL(mfma_f64_peak_kernel<2>, dim3(grid), dim3(256), 0, ctx->stream, its, sk); break;
      case 8: hipLaunchKernelGGL(mfma_f64_peak_kernel<8>, dim3(grid), dim3(256), 0, ctx->stream, its, sk); break;
      case 16: hipLaunchKernelGGL(mfma_f64_peak_kernel<16>, dim3(grid), dim3(256), 0, ctx->stream, its, sk); break;
      default: hipLaunchKernelGGL(mfma_f64_peak_kernel<4>, dim3(grid), dim3(256), 0, ctx->stream, its, sk); break;
    }
  };
  launch(n_it);  // warm-up (clock ramp)
  RT_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
  launch(n_it);
  RT_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
  RT_HIP_CHECK(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  RT_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)grid * 4 /*waves*/ * (double)n_it * nacc * 2048.0;
  *tflops = coissue ? (double)ms : flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return RT_OK;
}

int rt_bench_copy(rt_ctx* ctx, void* dst, const void* src, int64_t bytes, int reps, double* gbps) {
  if (!ctx || !dst || !src || !gbps || bytes < 32 || reps < 1) return RT_ERR_ARG;
  const long n4 = bytes / 32;
  hipEvent_t e0, e1;
  RT_HIP_CHECK(ctx, hipEventCreate(&e0));
  RT_HIP_CHECK(ctx, hipEventCreate(&e1));
  const int grid = ctx->num_cus * 8;
  hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<double4*>(dst),
                     static_cast<const double4*>(src), n4);
  RT_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<double4*>(dst),
                       static_cast<const double4*>(src), n4);
  RT_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
  RT_HIP_CHECK(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  RT_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
  *gbps = 2.0 * (double)(n4 * 32) * reps / (ms * 1e-3) / 1e9;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return RT_OK;
}

}  // extern "C"
