// Host-side small dense helpers of libromtime_hip.so (plain C++: also built with g++ and sanitizers by the CPU tests).
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/romtime_hip.h"

// Number of modes `orth` keeps: tol != 0 -> energy < tol (strict); else num != 0 -> min(num, n); else sigma > 1e-7.
int rt_truncation_rank(const std::vector<double>& s, const std::vector<double>& energy, int64_t num, double tol);

// H c = theta S c for symmetric H and positive definite S (k x k row-major): C = eigenvectors as columns, theta descending.
// False if S is not positive definite.  H and S are not modified.
bool rt_small_generalised_eigh(std::vector<double>& H, std::vector<double>& S, int k, std::vector<double>& C,
                               std::vector<double>& theta);
