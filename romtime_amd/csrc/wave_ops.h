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

// Variants whose masked-off lanes keep their own value (op(x, x) == x): for min / max / argmax.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_keep_i32(int x) {
  return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, BANK_MASK, false);
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_keep_f64(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = dpp_keep_i32<CTRL, ROW_MASK, BANK_MASK>((int)b), hi = dpp_keep_i32<CTRL, ROW_MASK, BANK_MASK>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double wave_max(double x) {
  x = fmax(x, dpp_keep_f64<0xb1>(x));
  x = fmax(x, dpp_keep_f64<0x4e>(x));
  x = fmax(x, dpp_keep_f64<0x114>(x));
  x = fmax(x, dpp_keep_f64<0x118>(x));
  x = fmax(x, dpp_keep_f64<0x142, 0xa>(x));
  x = fmax(x, dpp_keep_f64<0x143, 0xc>(x));
  return read_lane(x, 63);
}

__device__ __forceinline__ double wave_min(double x) {
  x = fmin(x, dpp_keep_f64<0xb1>(x));
  x = fmin(x, dpp_keep_f64<0x4e>(x));
  x = fmin(x, dpp_keep_f64<0x114>(x));
  x = fmin(x, dpp_keep_f64<0x118>(x));
  x = fmin(x, dpp_keep_f64<0x142, 0xa>(x));
  x = fmin(x, dpp_keep_f64<0x143, 0xc>(x));
  return read_lane(x, 63);
}

__device__ __forceinline__ int wave_min_i32(int x) {
  x = min(x, dpp_keep_i32<0xb1>(x));
  x = min(x, dpp_keep_i32<0x4e>(x));
  x = min(x, dpp_keep_i32<0x114>(x));
  x = min(x, dpp_keep_i32<0x118>(x));
  x = min(x, dpp_keep_i32<0x142, 0xa>(x));
  x = min(x, dpp_keep_i32<0x143, 0xc>(x));
  return __builtin_amdgcn_readlane(x, 63);
}

// argmax over the wave with the order (value descending, index ascending); wave-uniform result
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ void argmax_step(double& v, int& i) {
  const double ov = dpp_keep_f64<CTRL, ROW_MASK, BANK_MASK>(v);
  const int oi = dpp_keep_i32<CTRL, ROW_MASK, BANK_MASK>(i);
  const bool take = ov > v || (ov == v && oi < i);
  v = take ? ov : v;
  i = take ? oi : i;
}
__device__ __forceinline__ void wave_argmax(double& v, int& i) {
  argmax_step<0xb1>(v, i);
  argmax_step<0x4e>(v, i);
  argmax_step<0x114>(v, i);
  argmax_step<0x118>(v, i);
  argmax_step<0x142, 0xa>(v, i);
  argmax_step<0x143, 0xc>(v, i);
  v = read_lane(v, 63);
  i = __builtin_amdgcn_readlane(i, 63);
}

}  // namespace rtw
