// Closed-form local assembly of the 1-D P1 operators at given entries, for a batch of states.
//
// The reference evaluates an operator at the (M)DEIM entries by assembling the FEniCS form on the cells that touch
// each entry, one (mu, t) at a time (fom/base.py:523-599 with the forms of fom/nonlinear.py:374-494; the hyper-reduced
// online loop calls it for every operator and every step, deim.py:429-433).  On a uniform interval mesh of nx cells
// scaled to [0, L(mu, t)] the P1 element integrals are closed forms in the cell size h = L / nx (src/romtime/testing/
// mock.py:30-85 uses the same ones for the linear operators), so a whole table  F[state][entry]  - a state being one
// (step, mu) pair, or one basis function for the state-dependent operator - is one kernel launch:
//
//   mass        h/6 [[2, 1], [1, 2]]                          -> (i,i): c 4h/6      (i,i+-1): c h/6
//   stiffness   1/h [[1,-1],[-1, 1]]                          -> (i,i): 2c/h        (i,i+-1): -c/h
//   convection  -1/2 [[-1, 1], [-1, 1]]   (-int u' v)          -> (i,i): 0           (i,i+1): -c/2     (i,i-1): +c/2
//   trilinear   int w u' v, w P1 with nodal values w_k:
//               a_e = (2 w_e + w_e+1)/6, b_e = (w_e + 2 w_e+1)/6 -> (i,i-1): -b_i-1   (i,i): b_i-1 - a_i  (i,i+1): a_i
//   load        int f v, f P1 with nodal values f_k             -> (i): h/6 (f_i-1 + 4 f_i + f_i+1)
//   load_p2     int f v, f P2 on every cell (vertex and midpoint values; what FEniCS integrates for an
//               Expression(..., degree=2): fom/heat.py:119, fom/base.py:452-495), exact for quadratic f:
//               per cell  int f phi_l = h/6 (f_l + 2 f_m),  int f phi_r = h/6 (2 f_m + f_r)   (Simpson, exact for
//               the cubic integrand)                           -> (i): h/3 (f_i-1/2 + f_i + f_i+1/2)
//               The heat problem's forcing (problems/mfp1.py:38-39, quadratic in x) and lifting vector
//               -(int dg_dt v + alpha grad_g int v') = -h dg_dt(x_i) at interior dofs (fom/heat.py:131-169: dg_dt is
//               linear in x, grad_g constant so its term cancels between the two cells of a dof) are this rule.
//
// Dirichlet rows (first and last dof) are identity rows / zero load entries, as DirichletBC.apply leaves them
// (fom/base.py:501-521, 536-546).  The nodal function of the last two kinds is either given per state (n_states x N_h) or
// a ramp  amp[state] * node / nx  (the lifting function of the piston problem, g = amp x / L), or - load_p2 only - a
// polynomial  a0 + a1 x + a2 x^2  in the physical coordinate x = node h (three coefficients per state).  For load_p2
// the nodal function has 2 nx + 1 values per state: vertex k at 2k, midpoint of cell k at 2k + 1.
#include "common.h"

namespace {

struct P1Params {
  int kind, state_mode;   // state_mode: 0 none, 1 nodal values per state, 2 ramp with an amplitude per state, 3 polynomial
  long nx, n_states, m;
  const long* rows;
  const long* cols;       // nullptr for RT_P1_LOAD
  const double* h;        // n_states
  const double* coef;     // n_states or nullptr (= 1)
  const double* state;    // mode 1: n_states x (nx + 1); mode 2: n_states amplitudes
  double* out;            // n_states x m
};

__device__ __forceinline__ double nodal(const P1Params& p, long s, long k) {
  if (p.state_mode == 1) return p.state[s * (p.nx + 1) + k];
  return p.state[s] * ((double)k / (double)p.nx);
}

// load_p2: value k of the P2 function of state s (k = 2 * vertex, odd k = cell midpoints; x = k h / 2)
__device__ __forceinline__ double nodal_p2(const P1Params& p, long s, long k, double h) {
  if (p.state_mode == 1) return p.state[s * (2 * p.nx + 1) + k];
  const double x = 0.5 * h * (double)k;
  const double* a = p.state + 3 * s;
  return fma(fma(a[2], x, a[1]), x, a[0]);
}

__global__ void p1_local_assembly_kernel(const P1Params p) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long s = blockIdx.y;
  if (e >= p.m) return;
  const long i = p.rows[e], j = p.cols ? p.cols[e] : i;
  const bool dirichlet = (i == 0) || (i == p.nx);
  const double h = p.h[s], c = p.coef ? p.coef[s] : 1.0;
  const long d = j - i;
  double v = 0.0;
  if (p.kind == RT_P1_LOAD) {
    if (!dirichlet) v = c * h / 6.0 * (nodal(p, s, i - 1) + 4.0 * nodal(p, s, i) + nodal(p, s, i + 1));
  } else if (p.kind == RT_P1_LOAD_P2) {
    if (!dirichlet) v = c * h / 3.0 * (nodal_p2(p, s, 2 * i - 1, h) + nodal_p2(p, s, 2 * i, h) + nodal_p2(p, s, 2 * i + 1, h));
  } else if (dirichlet) {
    v = (d == 0) ? 1.0 : 0.0;
  } else if (d >= -1 && d <= 1) {
    switch (p.kind) {
      case RT_P1_MASS: v = c * h * (d == 0 ? 4.0 / 6.0 : 1.0 / 6.0); break;
      case RT_P1_STIFFNESS: v = c / h * (d == 0 ? 2.0 : -1.0); break;
      case RT_P1_CONVECTION: v = c * (d == 0 ? 0.0 : (d == 1 ? -0.5 : 0.5)); break;
      default: {  // RT_P1_TRILINEAR
        const double wm = nodal(p, s, i - 1), w0 = nodal(p, s, i), wp = nodal(p, s, i + 1);
        const double b_prev = (wm + 2.0 * w0) / 6.0, a_here = (2.0 * w0 + wp) / 6.0;
        v = c * (d == -1 ? -b_prev : (d == 0 ? b_prev - a_here : a_here));
      }
    }
  }
  p.out[s * p.m + e] = v;
}

}  // namespace

extern "C" int rt_p1_local_assembly(rt_ctx* ctx, int kind, int64_t nx, const int64_t* rows, const int64_t* cols, int64_t m,
                                    int64_t n_states, const double* h, const double* coef, int state_mode,
                                    const double* state, double* out) {
  if (!ctx) return RT_ERR_ARG;
  RT_ARG_CHECK(ctx, kind >= RT_P1_MASS && kind <= RT_P1_LOAD_P2 && nx >= 2 && rows && m >= 1 && n_states >= 1 && h && out);
  RT_ARG_CHECK(ctx, (kind == RT_P1_LOAD || kind == RT_P1_LOAD_P2) || cols);
  RT_ARG_CHECK(ctx, state_mode >= 0 && state_mode <= 3 && (state_mode == 0 || state));
  RT_ARG_CHECK(ctx, !(kind >= RT_P1_TRILINEAR && state_mode == 0));   // these integrate a nodal function
  RT_ARG_CHECK(ctx, (kind == RT_P1_LOAD_P2) ? (state_mode == 1 || state_mode == 3) : state_mode != 3);
  RT_ARG_CHECK(ctx, n_states <= 65535L * 65535L);
  P1Params p{kind, state_mode, (long)nx, (long)n_states, (long)m, reinterpret_cast<const long*>(rows),
             reinterpret_cast<const long*>(cols), h, coef, state, out};
  // states along grid.y in slabs of 65535
  for (long s0 = 0; s0 < n_states; s0 += 65535) {
    P1Params q = p;
    const long ns = (n_states - s0 < 65535) ? n_states - s0 : 65535;
    q.h = h + s0;
    q.coef = coef ? coef + s0 : nullptr;
    const long per_state = state_mode == 1 ? (kind == RT_P1_LOAD_P2 ? 2 * nx + 1 : nx + 1) : (state_mode == 3 ? 3 : 1);
    q.state = state ? state + s0 * per_state : nullptr;
    q.out = out + s0 * m;
    q.n_states = ns;
    hipLaunchKernelGGL(p1_local_assembly_kernel, dim3((unsigned)((m + 127) / 128), (unsigned)ns), dim3(128), 0, ctx->stream, q);
  }
  RT_HIP_CHECK(ctx, hipGetLastError());
  return RT_OK;
}
