// Closing a step of the hyper-reduced sweep and preparing the coefficient rows of the next one, for ONE parameter
// point: what hsweep_advance_kernel does per workgroup.  A header because the same code runs as the tail of the two
// solver kernels (solve.hip) - the step is then gemm + solve (+ the LU kernel for the systems the inverse tracking
// gave up on) instead of four launches, and the host of the pool's boxes sustains only about one launch per 20 us.
#pragma once

struct rt_advance {
  double* un;            // B x r   u^n            (updated)
  double* unm1;          // B x r   u^{n-1}        (updated when keep_prev)
  double* out;           // B x nt x r trajectory
  long step_done, nt;    // the step being closed; with ctr: taken from the device counter
  int keep_prev, do_coef;
  const double* Fm;      // tables of the NEXT step (or their bases, with ctr)
  const double* Fl;
  const double* W;
  const double* Cn;
  const double* Sn;
  int extrapolate, mm, ml, mn;
  double bdf, dt;
  double* G;             // 2 B x (mm + ml + mn) coefficient rows of the next step
  const long* ctr;       // graph replay: device step counter (or nullptr)
  int B;
  double* xT;            // r x B: the new u^n transposed as well (direct sweep: operand of the lift), or nullptr
  int enabled;           // 0: the solver kernels leave the step alone
};

// x: the step's solution for parameter point b (r values, global memory; visible to every thread of the workgroup);
// su: r doubles of LDS; t / nthreads: this thread and the workgroup's size.  Ends without a barrier.
__device__ __forceinline__ void hsweep_advance_rows(rt_advance a, int b, int r, const double* x, int do_store,
                                                    double* su, int t, int nthreads) {
  const int M = a.mm + a.ml + a.mn;
  if (a.ctr) {
    // graph replay: the step comes from the device counter; Fm / Fl / Cn / Sn are the table bases, bdf is that of
    // every step after the first
    a.step_done = *a.ctr;
    const long next = a.step_done + 1;
    a.do_coef = next < a.nt;
    const long s2 = a.do_coef ? next : 0;
    a.Fm += s2 * a.B * a.mm;
    if (a.Fl) a.Fl += s2 * a.B * a.ml;
    if (a.Cn) a.Cn += s2 * a.B * a.mn;
    if (a.Sn) a.Sn += s2 * a.B;
  }
  for (int j = t; j < r; j += nthreads) {
    double u = a.un[(long)b * r + j], up = a.unm1[(long)b * r + j];
    if (do_store) {
      const double v = x[(long)b * r + j];
      if (a.keep_prev) {
        a.unm1[(long)b * r + j] = u;
        up = u;
      }
      a.un[(long)b * r + j] = v;
      if (a.xT) a.xT[(long)j * a.B + b] = v;
      a.out[((long)b * a.nt + a.step_done) * r + j] = v;
      u = v;
    }
    su[j] = a.extrapolate ? 2.0 * u - up : u;
  }
  if (!a.do_coef) return;
  __syncthreads();
  const double sc = a.Sn ? a.Sn[b] : 1.0;
  for (int e = t; e < M; e += nthreads) {
    double g;
    if (e < a.mm) {
      g = a.bdf * a.Fm[(long)b * a.mm + e];
    } else if (e < a.mm + a.ml) {
      g = a.dt * a.Fl[(long)b * a.ml + (e - a.mm)];
    } else {
      const int q = e - a.mm - a.ml;
      double acc = a.Cn ? a.Cn[(long)b * a.mn + q] : 0.0;
      const double* w = a.W + (long)q * r;
      for (int j = 0; j < r; ++j) acc = fma(w[j], su[j], acc);
      g = a.dt * sc * acc;
    }
    a.G[(long)b * M + e] = g;
    a.G[(long)(a.B + b) * M + e] = (e < a.mm) ? a.Fm[(long)b * a.mm + e] : 0.0;
  }
}
