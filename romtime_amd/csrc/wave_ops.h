// Wave-wide reductions on the VALU data-parallel-primitive (DPP) path.  HIP's __shfl_* lower to
// ds_bpermute_b32 (an LDS-crossbar round trip per 32-bit half, ~100 cycles each and serial in a butterfly);
// the row-shift / row-broadcast DPP modifiers of gfx9 move data between lanes inside the VALU, so a 64-lane
// FP64 reduction costs six short steps.  Results are returned wave-uniform (read from lane 63 into SGPRs).
#pragma once
#include <hip/hip_runtime.h>

namespace rtw {

// lanes whose source is outside the row / masked off receive 0 (the identity of the sums below)
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double x) {
  const long long b = __double_as_longlong(x);
  int lo = (int)b, hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double read_lane(double x, int lane) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double first_lane(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// Sum over the 64 lanes (all must be active); the same value, bit for bit, in every lane.
__device__ __forceinline__ double wave_sum(double x) {
  x += dpp_f64<0xb1>(x);        // quad_perm:[1,0,3,2]
  x += dpp_f64<0x4e>(x);        // quad_perm:[2,3,0,1]
  x += dpp_f64<0x114>(x);       // row_shr:4
  x += dpp_f64<0x118>(x);       // row_shr:8   -> lane 15 of every row holds the row total
  x += dpp_f64<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
  x += dpp_f64<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return read_lane(x, 63);
}

}  // namespace rtw
