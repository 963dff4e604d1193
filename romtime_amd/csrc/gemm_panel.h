// Operand-panel staging shared by the FP64 MFMA kernels (gemm_mfma.hip, gram_mfma.hip).
#pragma once
#include "common.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

namespace rtk {

constexpr int KB = 16;            // contraction depth per LDS stage (4 MFMA k-steps)
constexpr int NTHREADS = 256;
constexpr int KSTRIDE = KB + 2;   // [m][k] image: 18-double rows -> conflict-free ds_read_b64

struct GemmParams {
  const double* A; long a_ks, a_ms;
  const double* B; long b_ks, b_ns;
  double* C; long c_rs, c_cs, c_split_stride;
  long K, M, Nn, k_per_split;
  int tiles_m, tiles_n, ntiles, splits, symmetric, vecA, vecB;
  double c_alpha, c_beta;  // direct (unsplit) store: C = c_alpha * acc (+ c_beta * C when c_beta != 0)
};

// One operand panel: KB (contraction) x BT (tile extent).
//   KC == false: the tile axis is contiguous in memory -> LDS image [k][m], row stride BT+16
//   KC == true : the contraction axis is contiguous   -> LDS image [m][k], row stride 18
// Both images give each half-wave of a ds_read_b64 (lane&15 -> m, lane>>4 -> k) 32 distinct
// 8-byte bank pairs.
template <int BT, bool KC, int NTHR = NTHREADS>
struct Panel {
  static constexpr int NL = (KB * BT / 2) / NTHR;
  static constexpr int SM = BT + 16;
  static constexpr int LDS = KC ? BT * KSTRIDE : KB * SM;

  static __device__ __forceinline__ void load(d2 (&regs)[NL], const double* __restrict__ P, long ks, long ms,
                                              long k0, long kend, long m0, long Mext, int vec, int tid) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int q = tid + NTHR * j;
      int kk, i;
      const double* p;
      bool v0, v1;
      if (!KC) {
        kk = q / (BT / 2);
        i = (q % (BT / 2)) * 2;
        const long k = k0 + kk, m = m0 + i;
        p = P + k * ks + m * ms;
        v0 = (k < kend) && (m < Mext);
        v1 = (k < kend) && (m + 1 < Mext);
        if (vec && v1) {
          regs[j] = *reinterpret_cast<const d2*>(p);
        } else {
          regs[j].x = v0 ? p[0] : 0.0;
          regs[j].y = v1 ? p[ms] : 0.0;
        }
      } else {
        i = q / (KB / 2);
        kk = (q % (KB / 2)) * 2;
        const long k = k0 + kk, m = m0 + i;
        p = P + m * ms + k * ks;
        v0 = (m < Mext) && (k < kend);
        v1 = (m < Mext) && (k + 1 < kend);
        if (vec && v1) {
          regs[j] = *reinterpret_cast<const d2*>(p);
        } else {
          regs[j].x = v0 ? p[0] : 0.0;
          regs[j].y = v1 ? p[ks] : 0.0;
        }
      }
    }
  }

  // interior panel (all KB rows and BT columns in range, 16-byte aligned pairs): no predicates, no branches
  static __device__ __forceinline__ void load_full(d2 (&regs)[NL], const double* __restrict__ P, long ks, long ms,
                                                   long k0, long m0, int tid) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int q = tid + NTHR * j;
      if (!KC) {
        const int kk = q / (BT / 2), i = (q % (BT / 2)) * 2;
        regs[j] = *reinterpret_cast<const d2*>(P + (k0 + kk) * ks + (m0 + i));
      } else {
        const int i = q / (KB / 2), kk = (q % (KB / 2)) * 2;
        regs[j] = *reinterpret_cast<const d2*>(P + (m0 + i) * ms + (k0 + kk));
      }
    }
  }

  // The same for KC == false with the address split the way the hardware wants it: a wave-uniform base (SGPR pair,
  // advanced by scalar adds from stage to stage) plus ONE 32-bit per-thread byte offset that never changes
  // (global_load_dwordx4 v, v_off, s[base]).  An FP64 MFMA blocks the VALU of its SIMD, so the 64-bit multiplies and
  // adds of a per-stage address computation (about 20 VALU instructions per stage and wave) cost matrix-core time.
  //   KC == false:  voff = ((tid / (BT/2)) * ks + (tid % (BT/2)) * 2) * 8,  step = (NTHR / (BT/2)) * ks * 8  (rows)
  //   KC == true :  voff = ((tid / (KB/2)) * ms + (tid % (KB/2)) * 2) * 8,  step = (NTHR / (KB/2)) * ms * 8  (tile columns)
  static __device__ __forceinline__ unsigned lane_offset(long ks, long ms, int tid) {
    return KC ? (unsigned)(((long)(tid / (KB / 2)) * ms + (tid % (KB / 2)) * 2) * 8)
              : (unsigned)(((long)(tid / (BT / 2)) * ks + (tid % (BT / 2)) * 2) * 8);
  }
  static __device__ __forceinline__ long load_step(long ks, long ms) {
    return KC ? (long)(NTHR / (KB / 2)) * ms * 8 : (long)(NTHR / (BT / 2)) * ks * 8;
  }
  static __device__ __forceinline__ void load_full_u(d2 (&regs)[NL], const char* __restrict__ base, unsigned voff,
                                                     long rows_step_bytes) {
#pragma unroll
    for (int j = 0; j < NL; ++j) regs[j] = *reinterpret_cast<const d2*>(base + j * rows_step_bytes + voff);
  }

  static __device__ __forceinline__ void store(const d2 (&regs)[NL], double* s, int tid) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int q = tid + NTHR * j;
      if (!KC) {
        const int kk = q / (BT / 2), i = (q % (BT / 2)) * 2;
        *reinterpret_cast<d2*>(&s[kk * SM + i]) = regs[j];
      } else {
        const int i = q / (KB / 2), kk = (q % (KB / 2)) * 2;
        *reinterpret_cast<d2*>(&s[i * KSTRIDE + kk]) = regs[j];
      }
    }
  }

  // MFMA operand of lane (l15 = lane&15 -> tile index, l4 = lane>>4 -> k) for k-step k4
  static __device__ __forceinline__ double frag(const double* s, int mloc, int k4, int l15, int l4) {
    return KC ? s[(mloc + l15) * KSTRIDE + k4 * 4 + l4] : s[(k4 * 4 + l4) * SM + mloc + l15];
  }
};

}  // namespace rtk
