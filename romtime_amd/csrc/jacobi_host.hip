// Host-side small dense step of the two-pass POD: cyclic two-sided Jacobi on a symmetric
// positive semi-definite k x k matrix with the relative stopping rule
//     |a_pq| <= eps * sqrt(a_pp a_qq)
// (Demmel & Veselic, "Jacobi's method is more accurate than QR", SIAM J. Matrix Anal. 1992).
// The second-pass Gram matrix G2 = Y^T Y of the rotated snapshots Y = X W1 is graded and
// nearly diagonal; this rule resolves its small eigenvalues to high RELATIVE accuracy, which
// LAPACK's dsyevd (absolute accuracy eps*||G2||) cannot, and that is what lets the Gram-based
// POD match dgesvd (pod.py:38) on modes far below sqrt(eps)*sigma_1.  k is the snapshot count
// (<= 512), so this is O(k^3) host work on a <= 2 MiB matrix, not a data-path fallback.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "common.h"

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
