// Host-side small dense step of the two-pass POD: cyclic two-sided Jacobi on a symmetric
// positive semi-definite k x k matrix with the relative stopping rule
//     |a_pq| <= eps * sqrt(a_pp a_qq)
// (Demmel & Veselic, "Jacobi's method is more accurate than QR", SIAM J. Matrix Anal. 1992).
// The second-pass Gram matrix G2 = Y^T Y of the rotated snapshots Y = X W1 is graded and
// nearly diagonal; this rule resolves its small eigenvalues to high RELATIVE accuracy, which
// LAPACK's dsyevd (absolute accuracy eps*||G2||) cannot, and that is what lets the Gram-based
// POD match dgesvd (pod.py:38) on modes far below sqrt(eps)*sigma_1.  k is the snapshot count
// (<= 512), so this is O(k^3) host work on a <= 2 MiB matrix, not a data-path fallback.
//
// This translation unit is plain C++ (no HIP header): the library build compiles it with hipcc like the rest, and
// tests/test_host_sanitizers.py compiles it a second time with g++ -fsanitize=address,undefined into a small driver
// (tests/host/host_dense_check.cpp) - the CPU-side sanitizer run SURVEY.md section 5 asks for; GPU sanitizers are not
// available on the pool.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "host_dense.h"

extern "C" int rt_host_jacobi_eigh(double* A, int64_t n, double* W, double* lam, int max_sweeps, int* sweeps_done) {
  if (!A || !W || !lam || n < 1) return RT_ERR_ARG;
  const double eps = 1.1102230246251565e-16;
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j) W[i * n + j] = (i == j) ? 1.0 : 0.0;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    long rotated = 0;
    for (int64_t p = 0; p + 1 < n; ++p) {
      for (int64_t q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        const double app = A[p * n + p], aqq = A[q * n + q];
        if (apq == 0.0 || std::fabs(apq) <= eps * std::sqrt(std::fabs(app) * std::fabs(aqq))) continue;
        ++rotated;
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        // rows p and q are contiguous: work on rows, then mirror (A stays symmetric)
        double* rp = A + p * n;
        double* rq = A + q * n;
        for (int64_t k = 0; k < n; ++k) {
          const double akp = rp[k], akq = rq[k];
          rp[k] = c * akp - s * akq;
          rq[k] = s * akp + c * akq;
        }
        for (int64_t k = 0; k < n; ++k) {
          A[k * n + p] = rp[k];
          A[k * n + q] = rq[k];
        }
        rp[p] = app - t * apq;
        rq[q] = aqq + t * apq;
        rp[q] = 0.0;
        rq[p] = 0.0;
        double* wp = W + p * n;  // W holds eigenvectors as ROWS while rotating (contiguous), transposed below
        double* wq = W + q * n;
        for (int64_t k = 0; k < n; ++k) {
          const double vp = wp[k], vq = wq[k];
          wp[k] = c * vp - s * vq;
          wq[k] = s * vp + c * vq;
        }
      }
    }
    if (rotated == 0) break;
  }
  if (sweeps_done) *sweeps_done = sweep;
  // sort descending; emit eigenvectors as COLUMNS of W
  std::vector<int64_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::vector<double> d(n);
  for (int64_t i = 0; i < n; ++i) d[i] = A[i * n + i];
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return d[a] > d[b]; });
  std::vector<double> rows(W, W + n * n);
  for (int64_t j = 0; j < n; ++j) {
    lam[j] = d[order[j]];
    for (int64_t k = 0; k < n; ++k) W[k * n + j] = rows[order[j] * n + k];
  }
  return RT_OK;
}

namespace {
constexpr double DROP_TOLERANCE = 1e-7;   // pod.py:4 (the docstring says 1e-8; the code is 1e-7)
}

// Number of modes `orth` keeps, with the reference's precedence tol > num > DROP_TOLERANCE (pod.py:46-57).
int rt_truncation_rank(const std::vector<double>& s, const std::vector<double>& energy, int64_t num, double tol) {
  const int n = (int)s.size();
  int r = 0;
  if (tol != 0.0) {
    for (int i = 0; i < n; ++i) r += (energy[i] < tol);   // strict: the mode that crosses tol is excluded
  } else if (num != 0) {
    r = (int)std::min<int64_t>(num, n);
  } else {
    for (int i = 0; i < n; ++i) r += (s[i] > DROP_TOLERANCE);
  }
  return r;
}

// Symmetric-definite k x k problem H c = theta S c on the host: S = L L^T, Jacobi on L^-1 H L^-T, back-substitution.
// C (k x k row-major) receives the eigenvectors as columns, theta descending.  Returns false if S is not positive definite.
bool rt_small_generalised_eigh(std::vector<double>& H, std::vector<double>& S, int k, std::vector<double>& C,
                            std::vector<double>& theta) {
  std::vector<double> L((size_t)k * k, 0.0);
  for (int j = 0; j < k; ++j) {
    double d = S[(size_t)j * k + j];
    for (int p = 0; p < j; ++p) d -= L[(size_t)j * k + p] * L[(size_t)j * k + p];
    if (!(d > 0.0)) return false;
    const double ljj = std::sqrt(d);
    L[(size_t)j * k + j] = ljj;
    for (int i = j + 1; i < k; ++i) {
      double v = S[(size_t)i * k + j];
      for (int p = 0; p < j; ++p) v -= L[(size_t)i * k + p] * L[(size_t)j * k + p];
      L[(size_t)i * k + j] = v / ljj;
    }
  }
  // B = L^-1 H L^-T: forward substitution on the rows, then on the columns
  std::vector<double> B(H);
  for (int c = 0; c < k; ++c)
    for (int i = 0; i < k; ++i) {
      double v = B[(size_t)i * k + c];
      for (int p = 0; p < i; ++p) v -= L[(size_t)i * k + p] * B[(size_t)p * k + c];
      B[(size_t)i * k + c] = v / L[(size_t)i * k + i];
    }
  for (int r = 0; r < k; ++r)
    for (int j = 0; j < k; ++j) {
      double v = B[(size_t)r * k + j];
      for (int p = 0; p < j; ++p) v -= L[(size_t)j * k + p] * B[(size_t)r * k + p];
      B[(size_t)r * k + j] = v / L[(size_t)j * k + j];
    }
  for (int i = 0; i < k; ++i)
    for (int j = i + 1; j < k; ++j) B[(size_t)i * k + j] = B[(size_t)j * k + i] = 0.5 * (B[(size_t)i * k + j] + B[(size_t)j * k + i]);
  // the Jacobi routine is written for PSD matrices (relative stopping rule); a shift keeps the rule meaningful for any sign
  double shift = 0.0;
  for (int i = 0; i < k; ++i) {
    double rowsum = 0.0;
    for (int j = 0; j < k; ++j) rowsum += std::fabs(B[(size_t)i * k + j]);
    shift = std::max(shift, rowsum);
  }
  for (int i = 0; i < k; ++i) B[(size_t)i * k + i] += shift;
  std::vector<double> W((size_t)k * k);
  theta.assign(k, 0.0);
  int sweeps = 0;
  if (rt_host_jacobi_eigh(B.data(), k, W.data(), theta.data(), 60, &sweeps) != RT_OK) return false;
  for (int i = 0; i < k; ++i) theta[i] -= shift;
  // C = L^-T W  (back substitution per column)
  C.assign((size_t)k * k, 0.0);
  for (int c = 0; c < k; ++c)
    for (int i = k - 1; i >= 0; --i) {
      double v = W[(size_t)i * k + c];
      for (int p = i + 1; p < k; ++p) v -= L[(size_t)p * k + i] * C[(size_t)p * k + c];
      C[(size_t)i * k + c] = v / L[(size_t)i * k + i];
    }
  return true;
}

