// rt_pod_orth: the whole of `orth` (src/romtime/rom/pod.py:7-62) behind ONE C entry point, for hosts that bind the C ABI
// without the Python layer.  Same algorithm as romtime_amd/pod.py (which remains the fast path of the Python drop-in:
// it overlaps the eigenvalue fetch with the back-projection and knows about process groups):
//
//   G = X^T X (rt_gram) -> column norms, D^-1 G D^-1 (rt_gram_scale) -> all eigenvalues (rt_sym_eig_values; host Jacobi for
//   n < 3) -> sigma, energy, truncation rank with the reference's precedence tol > num > 1e-7 (pod.py:46-57) ->
//     shallow spectrum (sigma_r >= 1e-2 sigma_1): k eigenvectors (+ a k x k Rayleigh-Ritz step when kept eigenvalues
//       are closer than 1e-4 lambda_1), Q = X D^-1 W S^-1;
//     deep spectrum: deflated levels - accept the modes within 1e-2 of the current largest singular value, project them
//       out of a working copy of the snapshots twice, Gram + eigensolve again (DESIGN.md "POD accuracy").
// The small dense steps on the host (truncation rule, k x k generalised eigenproblem of the Rayleigh-Ritz step) are
// O(k^3) scalar work on kilobytes; everything of size N_h stays on the device.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"
#include "host_dense.h"

namespace {

constexpr double TWO_PASS_RATIO = 1e-2;   // one Gram pass resolves vectors to eps (sigma_1/sigma_i)^2
constexpr double RR_GAP = 1e-4;           // eigenvalue gap (relative to lambda_1) below which inverse iteration is not trusted
constexpr int MAX_LEVELS = 12;

// Zs[i][j] = Z[i][j] * rowscale[i] * colscale[j]   (n x k, row-major; either scale may be null)
__global__ void scale_rows_cols_kernel(const double* __restrict__ Z, int n, int k, const double* __restrict__ rowscale_inv,
                                       const double* __restrict__ colscale, double* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * k) return;
  const int i = (int)(idx / k), j = (int)(idx % k);
  double v = Z[idx];
  if (rowscale_inv) v /= rowscale_inv[i];
  if (colscale) v *= colscale[j];
  out[idx] = v;
}

// Zs[i][j] = Z[i][j] / colnorm[i] / sqrt(lam[j])  (0 where lam[j] <= 0): the n x k matrix D^-1 W S^-1 of the back-projection,
// straight from the device eigenvalues - nothing of it needs the host
__global__ void backproject_weights_kernel(const double* __restrict__ Z, int n, int k, const double* __restrict__ colnorm,
                                           const double* __restrict__ lam, double* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * k) return;
  const int i = (int)(idx / k), j = (int)(idx % k);
  const double l = lam[j];
  const double inv = l > 0.0 ? 1.0 / sqrt(l) : 0.0;
  double v = Z[idx];
  if (colnorm) v /= colnorm[i];
  out[idx] = v * inv;
}

// dst (N x n row-major) = src (N x n, strides ks / ms) * diag(1 / colnorm)   (colnorm null: plain copy)
__global__ void copy_scaled_kernel(const double* __restrict__ src, long rs, long cs, long N, int n,
                                   const double* __restrict__ colnorm, double* __restrict__ dst) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * n) return;
  const long i = idx / n;
  const int j = (int)(idx % n);
  const double v = src[i * rs + j * cs];
  dst[idx] = colnorm ? v / colnorm[j] : v;
}

struct Eig {                  // eigen-decomposition of one level's Gram matrix
  std::vector<double> lam;    // host, descending
  double* lam_d = nullptr;    // device copy (n), valid when on_device
  bool on_device = false;
  std::vector<double> W_host; // n x n eigenvectors (columns), host route only
};

}  // namespace

extern "C" int rt_pod_backproject_weights(rt_ctx* ctx, const double* Z, int64_t n, int64_t k, const double* colnorm,
                                          const double* lam, double* Zs) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, Z && lam && Zs && n >= 1 && k >= 1 && k <= n && n * k < (1LL << 31));
  hipLaunchKernelGGL(backproject_weights_kernel, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, ctx->stream, Z, (int)n,
                     (int)k, colnorm, lam, Zs);
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}

extern "C" int rt_pod_enqueue(rt_ctx* ctx, const double* X, int64_t n_rows, int64_t n_cols, int64_t ld, int layout, int64_t k,
                              int normalize, double* G, double* colnorm, double* lam, int* status2, double* Z, double* Zs,
                              double* Q) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, X && G && colnorm && lam && status2 && Z && Zs && Q);
  RT_ARG_CHECK(ctx, n_rows >= 1 && n_cols >= 3 && n_cols <= 1024 && k >= 1 && k <= n_cols);
  RT_ARG_CHECK(ctx, layout == RT_ROW_MAJOR || layout == RT_COL_MAJOR);
  RT_TRY(rt_gram(ctx, X, n_rows, n_cols, ld, layout, G));
  RT_TRY(rt_gram_scale(ctx, G, n_cols, colnorm, normalize, status2 + 1));
  RT_TRY(rt_sym_eig_values(ctx, G, n_cols, lam, status2));
  RT_TRY(rt_sym_eig_vectors(ctx, n_cols, k, lam, Z));
  RT_TRY(rt_pod_backproject_weights(ctx, Z, n_cols, k, normalize ? colnorm : nullptr, lam, Zs));
  return rt_gemm_nn(ctx, X, ld, layout, Zs, k, n_rows, n_cols, k, Q, k, RT_ROW_MAJOR);
}

extern "C" int rt_pod_orth(rt_ctx* ctx, const double* X, int64_t n_rows, int64_t n_cols, int64_t ld, int layout, int64_t num,
                           double tol, int normalize, double* Q, int64_t q_cols, int64_t* r_out, double* s_host,
                           double* energy_host, int* levels_out) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, X && Q && r_out && s_host && energy_host && n_rows >= 1 && n_cols >= 1 && q_cols >= 0 && num >= 0);
  RT_ARG_CHECK(ctx, layout == RT_ROW_MAJOR || layout == RT_COL_MAJOR);
  RT_ARG_CHECK(ctx, ld >= (layout == RT_ROW_MAJOR ? n_cols : n_rows));
  if (n_cols > 1024) {
    ctx->err = "rt_pod_orth: more than 1024 snapshots (the device eigensolver's limit; use the pieces with a host eigensolver)";
    return RT_ERR_UNSUPPORTED;
  }
  const int n = (int)n_cols;
  const long N = n_rows;
  const long rs = (layout == RT_ROW_MAJOR) ? ld : 1, cs = (layout == RT_ROW_MAJOR) ? 1 : ld;
  hipStream_t st = ctx->stream;
  *r_out = 0;
  if (levels_out) *levels_out = 0;

  // small persistent device buffers of this call: hipMallocAsync keeps them off the ctx arenas, which the operators
  // called below carve for themselves
  struct DevBuf {
    hipStream_t st;
    std::vector<void*> ptrs;
    ~DevBuf() { for (void* p : ptrs) (void)hipFreeAsync(p, st); }
    double* get(size_t count) {
      void* p = nullptr;
      if (hipMallocAsync(&p, sizeof(double) * (count ? count : 1), st) != hipSuccess) return nullptr;
      ptrs.push_back(p);
      return static_cast<double*>(p);
    }
  } dev{st, {}};
  double* G = dev.get((size_t)n * n);
  double* colnorm = dev.get(n);
  double* lam_d = dev.get(n);
  double* Z = dev.get((size_t)n * n);       // eigenvectors n x k (k <= n)
  double* Zs = dev.get((size_t)n * n);
  double* scal = dev.get(n);                // inverse singular values of a level (k)
  double* small = dev.get((size_t)2 * n * n);  // H and S of the Rayleigh-Ritz step / the deflation coefficients (k x n)
  int* flags = reinterpret_cast<int*>(dev.get(2));
  if (!G || !colnorm || !lam_d || !Z || !Zs || !scal || !small || !flags) {
    ctx->err = "rt_pod_orth: hipMallocAsync failed";
    return RT_ERR_HIP;
  }

  auto eigensolve = [&](const double* Gm, Eig& e) -> int {
    e.lam.assign(n, 0.0);
    e.on_device = (n >= 3);
    if (e.on_device) {
      RT_TRY(rt_sym_eig_values(ctx, Gm, n, lam_d, flags + 1));
      int status = 0;
      RT_HIP_CHECK(ctx, hipMemcpyAsync(e.lam.data(), lam_d, sizeof(double) * n, hipMemcpyDeviceToHost, st));
      RT_HIP_CHECK(ctx, hipMemcpyAsync(&status, flags + 1, sizeof(int), hipMemcpyDeviceToHost, st));
      RT_HIP_CHECK(ctx, hipStreamSynchronize(st));
      e.lam_d = lam_d;
      if (status != 0 && ctx->eig_one_xcd) {  // hand-off timeout in the one-XCD form: once more in the general form
        ctx->eig_one_xcd = false;
        RT_TRY(rt_sym_eig_values(ctx, Gm, n, lam_d, flags + 1));
        RT_HIP_CHECK(ctx, hipMemcpyAsync(e.lam.data(), lam_d, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        RT_HIP_CHECK(ctx, hipMemcpyAsync(&status, flags + 1, sizeof(int), hipMemcpyDeviceToHost, st));
        RT_HIP_CHECK(ctx, hipStreamSynchronize(st));
      }
      if (status == 0) return RT_OK;
      e.on_device = false;  // timed out again: the host takes this eigenproblem
    }
    std::vector<double> A((size_t)n * n);
    RT_HIP_CHECK(ctx, hipMemcpyAsync(A.data(), Gm, sizeof(double) * n * n, hipMemcpyDeviceToHost, st));
    RT_HIP_CHECK(ctx, hipStreamSynchronize(st));
    e.W_host.assign((size_t)n * n, 0.0);
    int sweeps = 0;
    // shift to positive definiteness for the relative stopping rule (rounding can leave tiny negative eigenvalues)
    return rt_host_jacobi_eigh(A.data(), n, e.W_host.data(), e.lam.data(), 60, &sweeps);
  };

  // k leading eigenvectors of level `e` into Z (device, n x k row-major), Rayleigh-Ritz-repaired on G when clustered
  auto eigenvectors = [&](const double* Gm, Eig& e, int k) -> int {
    if (!e.on_device) {
      std::vector<double> Zh((size_t)n * k);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) Zh[(size_t)i * k + j] = e.W_host[(size_t)i * n + j];
      RT_HIP_CHECK(ctx, hipMemcpyAsync(Z, Zh.data(), sizeof(double) * n * k, hipMemcpyHostToDevice, st));
      RT_HIP_CHECK(ctx, hipStreamSynchronize(st));
      return RT_OK;
    }
    RT_TRY(rt_sym_eig_vectors(ctx, n, k, e.lam_d, Z));
    double gap = 1e300;
    for (int i = 0; i < k; ++i) gap = std::min(gap, e.lam[i] - (i + 1 < n ? e.lam[i + 1] : 0.0));
    if (gap >= RR_GAP * std::max(e.lam[0], 1e-300)) return RT_OK;
    // clustered: H = Z^T G Z, S = Z^T Z, H c = theta S c on the host, Z <- Z C
    double* GZ = Zs;  // free at this point
    RT_TRY(rt_gemm_nn(ctx, Gm, n, RT_ROW_MAJOR, Z, k, n, n, k, GZ, k, RT_ROW_MAJOR));
    RT_TRY(rt_gemm_tn(ctx, Z, k, RT_ROW_MAJOR, GZ, k, RT_ROW_MAJOR, n, k, k, small, k));
    RT_TRY(rt_gemm_tn(ctx, Z, k, RT_ROW_MAJOR, Z, k, RT_ROW_MAJOR, n, k, k, small + (size_t)k * k, k));
    std::vector<double> H((size_t)k * k), S((size_t)k * k), Cm, theta;
    RT_HIP_CHECK(ctx, hipMemcpyAsync(H.data(), small, sizeof(double) * k * k, hipMemcpyDeviceToHost, st));
    RT_HIP_CHECK(ctx, hipMemcpyAsync(S.data(), small + (size_t)k * k, sizeof(double) * k * k, hipMemcpyDeviceToHost, st));
    RT_HIP_CHECK(ctx, hipStreamSynchronize(st));
    for (int i = 0; i < k; ++i)
      for (int j = i + 1; j < k; ++j) {
        H[(size_t)i * k + j] = H[(size_t)j * k + i] = 0.5 * (H[(size_t)i * k + j] + H[(size_t)j * k + i]);
        S[(size_t)i * k + j] = S[(size_t)j * k + i] = 0.5 * (S[(size_t)i * k + j] + S[(size_t)j * k + i]);
      }
    if (!rt_small_generalised_eigh(H, S, k, Cm, theta)) {
      ctx->err = "rt_pod_orth: the Rayleigh-Ritz overlap matrix is not positive definite";
      return RT_ERR_HIP;
    }
    double worst = 0.0;
    for (int i = 0; i < k; ++i) worst = std::max(worst, std::fabs(theta[i] - e.lam[i]));
    if (worst > 1e-9 * std::max(e.lam[0], 1e-300)) {
      ctx->err = "rt_pod_orth: device eigenvectors failed the Rayleigh-Ritz cross-check";
      return RT_ERR_HIP;
    }
    RT_HIP_CHECK(ctx, hipMemcpyAsync(small, Cm.data(), sizeof(double) * k * k, hipMemcpyHostToDevice, st));
    RT_TRY(rt_gemm_nn(ctx, Z, k, RT_ROW_MAJOR, small, k, n, k, k, GZ, k, RT_ROW_MAJOR));
    RT_HIP_CHECK(ctx, hipMemcpyAsync(Z, GZ, sizeof(double) * n * k, hipMemcpyDeviceToDevice, st));
    RT_HIP_CHECK(ctx, hipStreamSynchronize(st));   // Cm is on this stack frame
    return RT_OK;
  };

  auto finish = [&](const std::vector<double>& s, const std::vector<double>& energy, int r) {
    // more snapshots than DoFs: the thin SVD has only min(N, n) singular values (pod.py:38)
    const int len = (int)std::min<long>(N, n);
    for (int i = 0; i < len; ++i) { s_host[i] = s[i]; energy_host[i] = energy[i]; }
    *r_out = std::min(r, len);
  };

  // ---- level 0 ------------------------------------------------------------------------------------------------------
  RT_TRY(rt_gram(ctx, X, n_rows, n_cols, ld, layout, G));
  RT_TRY(rt_gram_scale(ctx, G, n, colnorm, normalize, flags));
  Eig eig;
  RT_TRY(eigensolve(G, eig));
  int zero_norm = 0;
  RT_HIP_CHECK(ctx, hipMemcpyAsync(&zero_norm, flags, sizeof(int), hipMemcpyDeviceToHost, st));
  RT_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (normalize && zero_norm) {
    // the reference divides by a zero norm and scipy.linalg.svd then rejects the NaNs (pod.py:32-38)
    ctx->err = "rt_pod_orth: zero-norm snapshot with normalize (the reference raises: array must not contain infs or NaNs)";
    return RT_WARN_ZERO_NORM;
  }
  std::vector<double> s(n), energy(n);
  double total = 0.0;
  for (int i = 0; i < n; ++i) { s[i] = std::sqrt(std::max(eig.lam[i], 0.0)); total += s[i] * s[i]; }
  {
    double run = 0.0;
    for (int i = 0; i < n; ++i) { run += s[i] * s[i]; energy[i] = run / total; }   // NaN for an all-zero matrix, as the reference
  }
  int r = rt_truncation_rank(s, energy, num, tol);
  if (levels_out) *levels_out = 1;
  if (r == 0) {
    finish(s, energy, 0);
    return RT_OK;
  }
  if (std::min<long>(r, std::min<long>(N, n)) > q_cols) {
    finish(s, energy, r);
    ctx->err = "rt_pod_orth: Q has fewer columns than the modes kept (r_out holds the number needed)";
    return RT_ERR_ARG;
  }
  auto back_project = [&](const double* src, long src_rs, long src_cs, bool scale_rows, const std::vector<double>& sig, int k,
                          double* Qdst) -> int {
    std::vector<double> inv(k);
    for (int i = 0; i < k; ++i) inv[i] = sig[i] > 0.0 ? 1.0 / sig[i] : 0.0;
    RT_HIP_CHECK(ctx, hipMemcpyAsync(scal, inv.data(), sizeof(double) * k, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(scale_rows_cols_kernel, dim3((unsigned)(((long)n * k + 255) / 256)), dim3(256), 0, st, Z, n, k,
                       scale_rows ? colnorm : nullptr, scal, Zs);
    RT_HIP_CHECK(ctx, hipGetLastError());
    RT_HIP_CHECK(ctx, hipStreamSynchronize(st));   // inv is on this stack frame
    const int src_layout = (src_cs == 1) ? RT_ROW_MAJOR : RT_COL_MAJOR;
    const long src_ld = (src_cs == 1) ? src_rs : src_cs;
    return rt_gemm_nn(ctx, src, src_ld, src_layout, Zs, k, N, n, k, Qdst, q_cols, RT_ROW_MAJOR);
  };

  const bool deep = s[0] > 0.0 && s[r - 1] < TWO_PASS_RATIO * s[0];
  if (!deep) {
    RT_TRY(eigenvectors(G, eig, r));
    std::vector<double> sig(s.begin(), s.begin() + r);
    RT_TRY(back_project(X, rs, cs, normalize != 0, sig, r, Q));
    finish(s, energy, r);
    return RT_OK;
  }

  // ---- deflated levels (romtime_amd/pod.py::_pod_deflated) ---------------------------------------------------------------
  // working copy of the snapshots (row-major).  Not from the ctx's composite arena: the eigensolver keeps its reflectors
  // there between rt_sym_eig_values and rt_sym_eig_vectors.
  double* Xc = dev.get((size_t)N * n);
  if (!Xc) { ctx->err = "rt_pod_orth: hipMallocAsync failed (working copy of the snapshots)"; return RT_ERR_HIP; }
  bool have_copy = false;
  std::vector<double> s_acc;
  const int cap = (num != 0 && tol == 0.0) ? (int)std::min<int64_t>(num, n) : n;
  int levels = 0;
  std::vector<double> s_full(n, 0.0), e_full(n, 0.0);
  const double* Gl = G;
  double* G2 = nullptr;
  RT_HIP_CHECK(ctx, hipMemsetAsync(Q, 0, sizeof(double) * (size_t)N * q_cols, st));  // columns beyond the numerical rank stay zero
  for (;;) {
    ++levels;
    std::vector<double> sig(n);
    for (int i = 0; i < n; ++i) sig[i] = std::sqrt(std::max(eig.lam[i], 0.0));
    const int have = (int)s_acc.size();
    const int room = std::min<int>(cap, (int)q_cols) - have;
    const double floor_sig = have ? n * 2.220446049250313e-16 * s_acc[0] : 0.0;
    int k = 0;
    if (sig[0] > floor_sig && room > 0) {
      int cnt = 0;
      for (int i = 0; i < n; ++i) cnt += (sig[i] >= TWO_PASS_RATIO * sig[0]);
      k = std::min(std::max(1, cnt), room);
    }
    if (k > 0) {
      RT_TRY(eigenvectors(Gl, eig, k));
      std::vector<double> sk(sig.begin(), sig.begin() + k);
      if (!have_copy) RT_TRY(back_project(X, rs, cs, normalize != 0, sk, k, Q + have));
      else RT_TRY(back_project(Xc, n, 1, false, sk, k, Q + have));
      s_acc.insert(s_acc.end(), sk.begin(), sk.end());
    }
    const int got = (int)s_acc.size();
    std::fill(s_full.begin(), s_full.end(), 0.0);
    for (int i = 0; i < got; ++i) s_full[i] = s_acc[i];
    int tail_n = 0;
    for (int i = k; i < n && got + tail_n < n; ++i, ++tail_n) s_full[got + tail_n] = sig[i];
    double run = 0.0;
    for (int i = 0; i < n; ++i) { run += s_full[i] * s_full[i]; e_full[i] = run / total; }
    r = rt_truncation_rank(s_full, e_full, num, tol);
    if (r > q_cols && std::min<long>(r, std::min<long>(N, n)) > q_cols) {
      finish(s_full, e_full, r);
      ctx->err = "rt_pod_orth: Q has fewer columns than the modes kept (r_out holds the number needed)";
      return RT_ERR_ARG;
    }
    const double tail0 = (tail_n > 0) ? sig[k] : 0.0;
    if (r <= got || k == 0 || got >= n || levels >= MAX_LEVELS || tail_n == 0 || !(tail0 > 0.0)) break;
    // deflate: X <- X - Q_l (Q_l^T X), twice; the first sweep of the first level reads the caller's snapshots and writes
    // the (column-normalised) working copy
    const double* Ql = Q + have;
    for (int sweep = 0; sweep < 2; ++sweep) {
      double* Cm = small;   // k x n
      if (!have_copy) {
        hipLaunchKernelGGL(copy_scaled_kernel, dim3((unsigned)((N * n + 255) / 256)), dim3(256), 0, st, X, rs, cs, N, n,
                           normalize ? colnorm : nullptr, Xc);
        RT_HIP_CHECK(ctx, hipGetLastError());
        have_copy = true;
      }
      RT_TRY(rt_gemm_tn(ctx, Ql, q_cols, RT_ROW_MAJOR, Xc, n, RT_ROW_MAJOR, N, k, n, Cm, n));
      if (k <= 64) {
        RT_TRY(rt_rank_update(ctx, Xc, n, nullptr, Ql, q_cols, Cm, n, N, k, n, -1.0, Xc, n));
      } else {
        RT_TRY(rt_gemm_nn_axpby(ctx, Ql, q_cols, RT_ROW_MAJOR, Cm, n, N, k, n, -1.0, 1.0, Xc, n, RT_ROW_MAJOR));
      }
    }
    if (!G2) G2 = dev.get((size_t)n * n);
    if (!G2) { ctx->err = "rt_pod_orth: hipMallocAsync failed"; return RT_ERR_HIP; }
    RT_TRY(rt_gram(ctx, Xc, N, n, n, RT_ROW_MAJOR, G2));
    Gl = G2;
    RT_TRY(eigensolve(G2, eig));
  }
  if (levels_out) *levels_out = levels;
  finish(s_full, e_full, r);
  return RT_OK;
}
