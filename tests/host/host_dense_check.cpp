// CPU-side sanitizer run of the library's host code (romtime_amd/csrc/host_dense.cpp): built by
// tests/test_host_sanitizers.py with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all  and executed; any heap /
// stack overrun, use of uninitialised-by-construction memory the sanitizers see, signed overflow or misaligned access
// aborts the run.  The numerical checks double as known-answer tests of the three helpers.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../../romtime_amd/csrc/host_dense.h"

static int fail(const char* what, double v) {
  std::printf("FAIL %s (%g)\n", what, v);
  return 1;
}

int main() {
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd(0.0, 1.0);
  for (int n : {1, 2, 3, 7, 32, 65}) {
    // graded PSD matrix A = B diag(10^-i) B^T
    std::vector<double> B((size_t)n * n), A((size_t)n * n, 0.0), A0;
    for (auto& x : B) x = nd(rng);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += B[(size_t)i * n + k] * std::pow(10.0, -0.5 * k) * B[(size_t)j * n + k];
        A[(size_t)i * n + j] = s;
      }
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) A[(size_t)j * n + i] = A[(size_t)i * n + j];
    A0 = A;
    std::vector<double> W((size_t)n * n), lam(n);
    int sweeps = -1;
    if (rt_host_jacobi_eigh(A.data(), n, W.data(), lam.data(), 60, &sweeps) != RT_OK) return fail("jacobi rc", n);
    double res = 0.0, orth = 0.0, scale = std::fabs(lam[0]) + 1e-300;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double av = 0.0, ww = 0.0;
        for (int k = 0; k < n; ++k) { av += A0[(size_t)i * n + k] * W[(size_t)k * n + j]; ww += W[(size_t)k * n + i] * W[(size_t)k * n + j]; }
        res = std::fmax(res, std::fabs(av - lam[j] * W[(size_t)i * n + j]));
        orth = std::fmax(orth, std::fabs(ww - (i == j ? 1.0 : 0.0)));
      }
    if (res > 1e-12 * scale) return fail("jacobi residual", res);
    if (orth > 1e-12) return fail("jacobi orthogonality", orth);
    for (int i = 0; i + 1 < n; ++i)
      if (lam[i] < lam[i + 1]) return fail("jacobi order", i);
    // generalised problem with S = I + small SPD perturbation
    std::vector<double> H(A0), S((size_t)n * n, 0.0), C, theta;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) S[(size_t)i * n + j] = (i == j ? 1.0 : 0.0) + 1e-3 * A0[(size_t)i * n + j] / scale;
    if (!rt_small_generalised_eigh(H, S, n, C, theta)) return fail("generalised eigh rc", n);
    double gres = 0.0;
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        double hc = 0.0, sc = 0.0;
        for (int k = 0; k < n; ++k) { hc += H[(size_t)i * n + k] * C[(size_t)k * n + j]; sc += S[(size_t)i * n + k] * C[(size_t)k * n + j]; }
        gres = std::fmax(gres, std::fabs(hc - theta[j] * sc));
      }
    if (gres > 1e-11 * scale) return fail("generalised residual", gres);
    std::vector<double> bad(S);
    bad[0] = -1.0;
    if (rt_small_generalised_eigh(H, bad, n, C, theta)) return fail("indefinite S accepted", n);
  }
  if (rt_host_jacobi_eigh(nullptr, 3, nullptr, nullptr, 1, nullptr) != RT_ERR_ARG) return fail("null arguments", 0);
  // truncation rule (pod.py:46-57): tol (strict) > num > drop tolerance
  std::vector<double> s{3.0, 1.0, 1e-3, 5e-8, 0.0}, e(5);
  double tot = 0.0, run = 0.0;
  for (double v : s) tot += v * v;
  for (int i = 0; i < 5; ++i) { run += s[i] * s[i]; e[i] = run / tot; }
  if (rt_truncation_rank(s, e, 0, 0.0) != 3) return fail("drop rule", rt_truncation_rank(s, e, 0, 0.0));
  if (rt_truncation_rank(s, e, 2, 0.0) != 2 || rt_truncation_rank(s, e, 9, 0.0) != 5) return fail("num rule", 0);
  if (rt_truncation_rank(s, e, 2, 0.95) != 1) return fail("tol precedence / strictness", rt_truncation_rank(s, e, 2, 0.95));
  std::printf("host_dense_check ok\n");
  return 0;
}
