// CSR x dense and the reduced projection A_N = V^T (A V)  (src/romtime/utils.py:96-113).
//
// SpMM: one wavefront per matrix row; the 64 lanes stride the r columns of V, so every
// V-row read is one coalesced 512-B request and the 5-ish nonzeros of an FE row are a short
// scalar loop.  It is HBM/L2-bound (V rows are shared by neighbouring matrix rows and stay
// in L2); the dense contraction V^T (AV) then runs on the FP64 matrix cores (gemm_mfma.hip).
// The batched form serves MDEIM.project_basis (mdeim.py:153-192): the B value-vectors share
// one pattern, so AV for a chunk of modes is laid side by side ([N][chunk*r]) and a single
// wide V^T [AV_0 | AV_1 | ...] GEMM produces all r x r blocks of the chunk.
#include "common.h"

namespace {

// Y[row][bb*r + c] = sum_e data[bb][e] * V[indices[e]][c]   for bb < nb (nb value vectors, one pattern)
__global__ __launch_bounds__(256) void csr_spmm_kernel(const long* __restrict__ indptr,
                                                       const long* __restrict__ indices,
                                                       const double* __restrict__ data, long d_es, long d_bs,
                                                       int nb, long N, const double* __restrict__ V, long ldv,
                                                       int r, double* __restrict__ Y, long ldy) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long row = wave; row < N; row += nwaves) {
    const long e0 = indptr[row], e1 = indptr[row + 1];
    for (int bb = 0; bb < nb; ++bb) {
      for (int c = lane; c < r; c += 64) {
        double acc = 0.0;
        for (long e = e0; e < e1; ++e) acc = fma(data[e * d_es + bb * d_bs], V[indices[e] * ldv + c], acc);
        Y[row * ldy + (long)bb * r + c] = acc;
      }
    }
  }
}

// AN_batch[b][i][j] = W[i][b*r + j]   (W is r x (nb*r) row-major)
__global__ void unpack_blocks_kernel(const double* __restrict__ W, int nb, int r, double* __restrict__ AN) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)nb * r * r;
  if (idx >= total) return;
  const int b = (int)(idx / ((long)r * r));
  const int rem = (int)(idx % ((long)r * r));
  const int i = rem / r, j = rem % r;
  AN[idx] = W[(long)i * nb * r + (long)b * r + j];
}

int spmm_launch(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, const double* data, long d_es,
                long d_bs, int nb, int64_t N, const double* V, int64_t ldv, int64_t r, double* Y, int64_t ldy) {
  long blocks = (N + 3) / 4;
  const long cap = (long)ctx->num_cus * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(csr_spmm_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const long*>(indptr), reinterpret_cast<const long*>(indices), data, d_es, d_bs,
                     nb, (long)N, V, (long)ldv, (int)r, Y, (long)ldy);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

}  // namespace

extern "C" int rt_csr_spmm(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, const double* data,
                           int64_t N, const double* V, int64_t ldv, int64_t r, double* Y, int64_t ldy) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, indptr && indices && data && V && Y && N >= 1 && r >= 1 && ldv >= r && ldy >= r);
  return spmm_launch(ctx, indptr, indices, data, 1, 0, 1, N, V, ldv, r, Y, ldy);
}

extern "C" int rt_project_csr(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, const double* data,
                              int64_t N, const double* V, int64_t ldv, int64_t r, double* AN) {
  return rt_project_csr_batched(ctx, indptr, indices, data, 0, RT_COL_MAJOR, 1, N, V, ldv, r, AN);
}

extern "C" int rt_project_csr_batched(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices,
                                      const double* data_batch, int64_t ld_data, int data_layout, int64_t B,
                                      int64_t N, const double* V, int64_t ldv, int64_t r, double* AN_batch) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, indptr && indices && data_batch && V && AN_batch);
  RT_ARG_CHECK(ctx, N >= 1 && r >= 1 && B >= 1 && ldv >= r);
  RT_ARG_CHECK(ctx, data_layout == RT_ROW_MAJOR || data_layout == RT_COL_MAJOR);
  const long d_es = (data_layout == RT_COL_MAJOR) ? 1 : ld_data;  // stride between entries of one vector
  const long d_bs = (data_layout == RT_COL_MAJOR) ? ld_data : 1;  // stride between vectors

  {
    const int frc = rt_project_fused(ctx, indptr, indices, data_batch, d_es, d_bs, B, N, V, ldv, r, AN_batch);
    if (frc != RT_ERR_UNSUPPORTED) return frc;
  }
  // r > 128: unfused route.  Chunk the modes so that the AV workspace stays <= ~2 GiB
  long chunk = (long)((2ull << 30) / (sizeof(double) * (size_t)N * (size_t)r));
  if (chunk < 1) chunk = 1;
  if (chunk > B) chunk = B;
  // AV (N x chunk*r) | W (r x chunk*r) live in the composite arena; the GEMM's split slab
  // comes from the leaf arena (rt_scratch), so the two never alias.
  const size_t av_bytes = sizeof(double) * (size_t)N * (size_t)chunk * (size_t)r;
  const size_t w_bytes = sizeof(double) * (size_t)r * (size_t)chunk * (size_t)r;
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  void* base = nullptr;
  int rc = rt_scratch2(ctx, up(av_bytes) + up(w_bytes), &base);
  if (rc != RT_OK) return rc;
  double* AV = static_cast<double*>(base);
  double* W = reinterpret_cast<double*>(static_cast<char*>(base) + up(av_bytes));

  for (long b0 = 0; b0 < B; b0 += chunk) {
    const int nb = (int)((B - b0 < chunk) ? (B - b0) : chunk);
    const long ldy = (long)nb * r;
    rc = spmm_launch(ctx, indptr, indices, data_batch + b0 * d_bs, d_es, d_bs, nb, N, V, ldv, r, AV, ldy);
    if (rc != RT_OK) return rc;
    if (nb == 1) {
      rc = rt_gemm_strided(ctx, V, ldv, 1, AV, ldy, 1, N, r, r, AN_batch + b0 * r * r, r, 1, false, true);
    } else {
      rc = rt_gemm_strided(ctx, V, ldv, 1, AV, ldy, 1, N, r, ldy, W, ldy, 1, false, true);
      if (rc == RT_OK) {
        const long total = (long)nb * r * r;
        hipLaunchKernelGGL(unpack_blocks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, W,
                           nb, (int)r, AN_batch + b0 * r * r);
      }
    }
    if (rc != RT_OK) return rc;
    RT_HIP_CHECK(ctx, hipGetLastError());
  }
  return RT_OK;
}
