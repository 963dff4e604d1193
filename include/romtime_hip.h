/*
 * romtime_hip.h -- C ABI of libromtime_hip.so (MI355X / gfx950).
 *
 * The reference (KikeM/romtime @ v0) has no FFI layer: its "operator API" is a
 * Python class surface over NumPy/SciPy calls.  Each entry point
 * below replaces one of those library call sites; the citation is the
 * reference file:line (relative to /root/reference) it stands in for.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer (hipMalloc'ed or a torch CUDA
 *     tensor's data_ptr()); f64 = double, indices = int64_t;
 *   - work is enqueued on the ctx's stream (rt_ctx_set_stream) and the call
 *     returns without synchronising unless stated;
 *   - return value: 0 ok, <0 argument / runtime error (rt_last_error has the
 *     text), >0 numerical warning (RT_WARN_*); nothing throws;
 *   - the caller owns all buffers; a ctx owns a scratch arena that grows on
 *     demand (hipMalloc: not capturable on first use);
 *   - one ctx per host thread; calls on one ctx are serialised by its stream.
 */
#ifndef ROMTIME_HIP_H
#define ROMTIME_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

#define RT_OK 0
#define RT_ERR_ARG (-1)
#define RT_ERR_HIP (-2)
#define RT_ERR_UNSUPPORTED (-3)
#define RT_WARN_ZERO_NORM 1   /* zero-norm column in orth(normalize=True), pod.py:32-33 yields NaN */
#define RT_WARN_SINGULAR 2    /* zero pivot in a dense solve */

#define RT_ROW_MAJOR 0        /* element (i,j) at p[i*ld + j]  (NumPy C order)  */
#define RT_COL_MAJOR 1        /* element (i,j) at p[j*ld + i]  (NumPy F order; np.array(list_of_vectors).T, deim.py:384) */

int rt_version(void);
int rt_ctx_create(rt_ctx** out, int device);
void rt_ctx_destroy(rt_ctx* ctx);
int rt_ctx_set_stream(rt_ctx* ctx, void* hip_stream);
int rt_ctx_synchronize(rt_ctx* ctx);
const char* rt_last_error(rt_ctx* ctx);
/* dispatch statistics of the most recent GEMM-class launch: [0]=grid, [1]=splits, [2]=tile */
int rt_last_launch_info(rt_ctx* ctx, int64_t* info3);

/* Measurement: when on, every GEMM-class call brackets its main MFMA kernel (not the slab
 * reduction) with hipEvents on the ctx stream; rt_last_gemm_ms synchronises on the second event
 * and returns that kernel's duration in milliseconds. */
int rt_ctx_set_profile(rt_ctx* ctx, int on);
/* Named switches.  "eig_one_xcd" (default 1): the small eigensolver may place its cooperating workgroups on one
 * XCD and hand off through that XCD's L2 (placement is verified on the device); it needs every CU of that XCD,
 * so processes that share a GPU turn it off (a hand-off timeout of rt_sym_eig_values is the symptom).
 * "eig_xcd" (default 0, 0 .. 7): the XCD that form puts its workgroups on.  Contexts with different values (each on
 * a stream of its own) run their eigensolves side by side - eight small PODs at a time, romtime_amd.pipeline.PodLanes;
 * two eigensolves in flight on the SAME XCD would starve each other of CUs (they end in the hand-off's time-out).
 * "sweep_graph" (default 0): rt_hrom_bdf_sweep captures one time step as a hipGraph (its kernels read the step from a
 * device counter) and replays it for steps 1 .. nt-1 instead of launching two kernels per step; for hosts that
 * cannot keep ahead of the device.  The call then returns only when the sweep has finished. */
int rt_ctx_set_option(rt_ctx* ctx, const char* name, int value);
/* "cu_limit" (default 0 = the device's CU count; a multiple of 8): the number of CUs this ctx sizes its persistent grids
 * for - set it on a ctx whose stream is CU-masked (below) so that, e.g., the snapshot Gram kernel launches exactly the
 * workgroups its share of the chip holds.
 * "gram_pace" (default 1): the workgroups of an XCD in the snapshot Gram kernel keep within two stages of each other
 * (a progress word per XCD, bounded naps for leaders), so that the tiles sharing a panel read it from the L2 while it is
 * still there: L2<-fabric reads of the off-diagonal launch 10.4 -> 4.13 GB on 1e6 x 512 at no cost in time on the whole
 * chip; 0 switches it off (the POD pipeline does, for the Gram on its 224-CU stream: +2 % there). */

/* CU-partitioned streams (throughput mode of the POD: the n x n eigensolve of one snapshot set runs beside the Gram
 * kernel of the next on disjoint CUs).  Creates a HIP stream whose kernels run only on CUs [first, first + count) of
 * EVERY XCD (MI355X: 32 CUs per XCD; a mask must leave no XCD empty).  *stream is a hipStream_t for rt_ctx_set_stream. */
int rt_stream_create_cu_range(int device, int first_cu_per_xcd, int n_cu_per_xcd, void** stream);
int rt_stream_destroy(void* stream);
int rt_last_gemm_ms(rt_ctx* ctx, double* ms);
/* Event counters kept on the device by the kernels themselves (nothing on the hot path waits for them); reading one
 * synchronises the ctx stream.  Names: "eig_timeouts" (hand-offs of the small eigensolver that hit their wall-clock
 * bound: results of that call were invalid and the caller took another route), "eig_one_xcd" / "eig_general_form"
 * (tridiagonalisations that ran each hand-off form), "gram_off_xcd" (workgroups of the snapshot Gram kernel that ran on
 * another XCD than the one their K range was laid out for: should stay 0), "sweep_newton_iterations", "sweep_restarts",
 * "sweep_lu_fallbacks", "sweep_solves" (the four numbers of rt_last_sweep_stats). */
int rt_ctx_get_counter(rt_ctx* ctx, const char* name, int64_t* value);
/* The same for the most recent launch of the snapshot Gram kernel (rt_gram, n >= 97, long X): its own event pair,
 * so it can be read at the end of a POD step, after the GEMMs that followed it, without holding the host back. */
int rt_last_gram_ms(rt_ctx* ctx, double* ms);
/* Which form rt_gram takes for an n_rows x n_cols snapshot set on num_cus compute units (host logic only, no GPU needed:
 * diagnostics and the CPU-side tests): out[0] = 0 the generic symmetric GEMM, 1 two launches (out[3] / out[4] sub-splits
 * per off-diagonal / diagonal tile and XCD), 2 one launch with uniform slots (out[1] / out[2] per tile) whose paced
 * workgroups read the snapshots once. */
int rt_gram_plan_info(int num_cus, int64_t n_rows, int64_t n_cols, int* out);

/* ---- POD (src/romtime/rom/pod.py:7-62) ------------------------------------------------ */

/* G = X^T X (n_cols x n_cols, row-major, full symmetric).  X is n_rows x n_cols with
 * leading dimension ld in the given layout.  This is the local part of the snapshot
 * Gram matrix: with row-sharded X the caller all-reduces G (RCCL) before rt_gram_scale.
 * Replaces the O(N n^2) part of scipy.linalg.svd(..., lapack_driver="gesvd") (pod.py:38). */
int rt_gram(rt_ctx* ctx, const double* X, int64_t n_rows, int64_t n_cols, int64_t ld, int layout,
            double* G);

/* colnorm[j] = sqrt(G[j][j]) (== np.linalg.norm(X, axis=0), pod.py:32); if normalize != 0,
 * G <- D^-1 G D^-1 (the Gram matrix of np.divide(X, l2_norms), pod.py:33).  A zero norm
 * propagates NaN exactly as the reference does and *status_flag (device int, may be NULL)
 * is set to RT_WARN_ZERO_NORM. */
int rt_gram_scale(rt_ctx* ctx, double* G, int64_t n, double* colnorm, int normalize, int* status_flag);

/* The whole of `orth` (pod.py:7-62) in one call, for hosts that bind this library without the Python layer: column
 * norms (normalize != 0), Gram matrix, all singular values, energy curve, truncation with the reference's precedence
 * (tol != 0: energy < tol, strict; else num != 0: first num; else sigma > 1e-7), basis.  X: n_rows x n_cols DEVICE matrix
 * (ld, layout), n_cols <= 1024.  Q: DEVICE buffer n_rows x q_cols row-major; on return its first *r_out columns are the
 * basis (the others are zero or unspecified); if the rule keeps more than q_cols modes the call returns RT_ERR_ARG with
 * *r_out = the number needed.  s_host / energy_host: HOST arrays of min(n_rows, n_cols) entries - ALL singular values and
 * the whole energy curve, as the reference returns them.  *levels_out (may be NULL): Gram passes taken (1 = single pass;
 * more = deflated levels for spectra deeper than 1e-2).  Returns RT_WARN_ZERO_NORM (and no basis) where
 * the reference raises on a zero-norm snapshot with normalize.  Synchronises the ctx stream (the truncation rule needs
 * the spectrum on the host).  Single device: a row-sharded host sums the Gram matrices itself and uses the pieces below. */
int rt_pod_orth(rt_ctx* ctx, const double* X, int64_t n_rows, int64_t n_cols, int64_t ld, int layout, int64_t num,
                double tol, int normalize, double* Q, int64_t q_cols, int64_t* r_out, double* s_host,
                double* energy_host, int* levels_out);

/* One single-pass POD, ENQUEUED on the ctx stream and not waited for: Gram matrix, scaling, all eigenvalues, the k
 * leading eigenvectors, back-projection Q (n_rows x k, row-major) = X D^-1 W S^-1.  The decisions `orth` takes from the
 * spectrum (deflated levels for deep spectra, Rayleigh-Ritz for clusters, the zero-norm error) are the caller's, after
 * the fact: lam (n_cols, descending, device), status2 (device ints: [0] eigensolver hand-off status, [1] zero-norm
 * flag), colnorm (n_cols).  G (n x n), Z and Zs (n x k) are work space the caller owns; 3 <= n_cols <= 1024, k <= n_cols.
 * One host call per snapshot set:
 * what lets romtime_amd.pipeline.PodLanes keep eight small PODs on the chip (eight contexts, "eig_xcd" 0 .. 7). */
int rt_pod_enqueue(rt_ctx* ctx, const double* X, int64_t n_rows, int64_t n_cols, int64_t ld, int layout, int64_t k,
                   int normalize, double* G, double* colnorm, double* lam, int* status2, double* Z, double* Zs, double* Q);

/* Zs (n x k row-major) = D^-1 W S^-1: W = Z (n x k eigenvectors of the Gram matrix, rt_sym_eig_vectors), D = diag(colnorm)
 * (NULL: identity, normalize == False), S_j = sqrt(lam_j) (device eigenvalues; S^-1 = 0 where lam_j <= 0).  The matrix the
 * back-projection Q = X Zs multiplies (pod.py:33,38 combined); one launch, no host data. */
int rt_pod_backproject_weights(rt_ctx* ctx, const double* Z, int64_t n, int64_t k, const double* colnorm, const double* lam,
                               double* Zs);

/* C (m x n, row-major, ldc) = A^T B with A: N x m, B: N x n (each with ld + layout).
 * np.matmul(V.T, AhV) (utils.py:112), np.matmul(V.T, Vfh) (deim.py:509), V.T.dot(f)
 * (rom.py:133,156).  A == B with m == n computes only the upper triangle and mirrors it. */
int rt_gemm_tn(rt_ctx* ctx, const double* A, int64_t lda, int a_layout, const double* B, int64_t ldb,
               int b_layout, int64_t N, int64_t m, int64_t n, double* C, int64_t ldc);

/* Y (N x k, ldy, y_layout) = X (N x n, ldx, x_layout) * T (n x k, row-major, ldt).
 * The POD back-projection U_r = X (D^-1 W_r S_r^-1) and V.dot(uN) (rom.py:111-112). */
int rt_gemm_nn(rt_ctx* ctx, const double* X, int64_t ldx, int x_layout, const double* T, int64_t ldt,
               int64_t N, int64_t n, int64_t k, double* Y, int64_t ldy, int y_layout);
/* Y = beta Y + alpha X T, same operands.  The deflation of the multi-level POD, X <- X - Q (Q^T X), in place
 * (Gram-Schmidt sweep between the levels; the reference gets deep spectra from dgesvd itself, pod.py:38). */
int rt_gemm_nn_axpby(rt_ctx* ctx, const double* X, int64_t ldx, int x_layout, const double* T, int64_t ldt,
                     int64_t N, int64_t n, int64_t k, double alpha, double beta, double* Y, int64_t ldy,
                     int y_layout);

/* Thin update of a tall row-major matrix: Y_dst (N x n) = Y_src diag(colscale) + alpha X (N x k) T (k x n), k <= 64
 * (RT_ERR_UNSUPPORTED above), colscale optional (NULL = 1), Y_src == Y_dst allowed.  The deflation sweep
 * X <- X - Q (Q^T X) of the levelled POD (romtime_amd/pod.py; replaces what dgesvd's bidiagonalisation does
 * implicitly in rom/pod.py:38); HBM-bound streaming kernel. */
int rt_rank_update(rt_ctx* ctx, const double* Ysrc, int64_t ldys, const double* colscale, const double* X, int64_t ldx,
                   const double* T, int64_t ldt, int64_t N, int64_t k, int64_t n, double alpha, double* Ydst,
                   int64_t ldyd);

/* Out-of-place transpose: dst (cols x rows, row-major, ld_dst) = src (rows x cols, row-major, ld_src)^T. */
int rt_transpose(rt_ctx* ctx, const double* src, int64_t rows, int64_t cols, int64_t ld_src, double* dst,
                 int64_t ld_dst);

/* ---- DEIM (src/romtime/deim/deim.py:517-561, :159, :212) -------------------------------- */

/* Greedy interpolation-index selection on the collateral basis Phi (N x m).
 * idx[k] = argmax_i |phi_k - Phi[:, :k] c|, c = solve(Phi[idx[:k], :k], phi_k[idx[:k]]),
 * first maximum on ties (np.argmax, deim.py:531,553).  PT_U (m x m row-major) = Phi[idx, :]
 * (np.matmul(P.T, basis), deim.py:159,212).  margin (m, may be NULL) = (top1-top2)/top1 of
 * |residual| per step.  Synchronises the stream before returning. */
int rt_deim_greedy(rt_ctx* ctx, const double* Phi, int64_t N, int64_t m, int64_t ld, int layout,
                   int64_t* idx, double* PT_U, double* margin);

/* ---- projections (src/romtime/utils.py:96-113,136-149; deim/mdeim.py:153-192) ----------- */

/* Y (N x r row-major, ldy) = A V, A in CSR (N rows), V (n_colsA x r row-major, ldv).
 * scipy.sparse.csr_matrix.dot(dense) (utils.py:111). */
int rt_csr_spmm(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, const double* data, int64_t N,
                const double* V, int64_t ldv, int64_t r, double* Y, int64_t ldy);

/* AN (r x r row-major) = V^T (A V)   (project_csr, utils.py:96-113). */
int rt_project_csr(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices, const double* data, int64_t N,
                   const double* V, int64_t ldv, int64_t r, double* AN);

/* Batched over B value-vectors on one fixed CSR pattern: data_batch holds B value vectors,
 * vector b at data_batch + b*ld_data (stride 1) when data_layout == RT_COL_MAJOR, or element
 * e of vector b at data_batch[e*ld_data + b] when RT_ROW_MAJOR (the (nnz x m) basis_fom array
 * in C order).  AN_batch is B x r x r.  MDEIM.project_basis (mdeim.py:153-192) with
 * vector_to_csr folded away (the pattern is already CSR-ordered, mdeim.py:145-149), and the
 * online multi-parameter sweep. */
int rt_project_csr_batched(rt_ctx* ctx, const int64_t* indptr, const int64_t* indices,
                           const double* data_batch, int64_t ld_data, int data_layout, int64_t B, int64_t N,
                           const double* V, int64_t ldv, int64_t r, double* AN_batch);

/* ---- reduced solve (np.linalg.solve: deim.py:491-492; gmres on a dense r x r: rom.py:492) --- */

/* Solve K_b x_b = rhs_b for b < B by LU with partial pivoting, one workgroup per system.
 * K (B x r x r row-major) is left untouched (the factors are not an output), rhs (B x r) is overwritten with x.
 * info (B device ints, may be NULL): 0, or RT_WARN_SINGULAR. r <= 128. */
int rt_dense_solve_batched(rt_ctx* ctx, double* K, double* rhs, int64_t r, int64_t B, int* info);
/* K X = B for many right-hand sides against ONE r x r matrix (r <= 128): K row-major (device, untouched), B and X
 * r x nrhs row-major (device, distinct), *info (device int, may be NULL) = RT_WARN_SINGULAR on an exactly zero pivot.
 * Pivoted LU in LDS, one thread per right-hand side.  Replaces np.linalg.solve(PT_U.T, basis_rom.T) - the theta solve of
 * deim.py:477-493 applied once to every column of a projected collateral basis, when the interpolation matrix is folded
 * into the expansion of a hyper-reduced operator (romtime_amd/sweep.py). */
int rt_dense_solve_multi(rt_ctx* ctx, const double* K, int64_t r, const double* B, double* X, int64_t nrhs, int* info);


/* The same solve for a SEQUENCE of slowly changing matrices (the reduced systems of consecutive time steps differ by
 * O(dt)): K_b^-1 is tracked in Xinv (B x r x r, caller-owned device memory carried from call to call) and refreshed by
 * Newton-Schulz iterations on the matrix cores, x = Xinv b plus one step of iterative refinement against K; the first
 * call (have_prev = 0), a matrix that moved too far, or one the iteration gives up on (singular to working precision:
 * then pivoted LU inside the same kernel) cost more, the answers agree with rt_dense_solve_batched to rounding.  This is
 * what rt_rom_bdf_sweep / rt_hrom_bdf_sweep call every step.  info as above.  r <= 80 (three padded matrices in LDS):
 * RT_ERR_UNSUPPORTED beyond.  rt_ctx_get_counter("sweep_lu_fallbacks" | "sweep_restarts" | ...) tells what happened. */
int rt_tracked_solve_batched(rt_ctx* ctx, const double* K, double* Xinv, double* rhs, int64_t r, int64_t B, int have_prev,
                             int* info);

/* ---- online sweep (RomConstructor*.solve, rom/rom.py:430-555, :877-929) on the device --------- */
typedef struct {
  int64_t N, nnz, r, n_mu, nt;   /* DoFs, pattern nonzeros, reduced size (<= 128), parameter points, time steps */
  double dt;
  int bdf2;                      /* 1: BDF2 with u* = 2 u_h - u_h^{n-1} (fom.BDF_SCHEME == "2"), 0: BDF1 */
  const int64_t* indptr;         /* CSR pattern shared by every operator (device) */
  const int64_t* indices;
  const double* V;               /* N x r row-major reduced basis */
  const double* mass_values;     /* nnz: mass matrix values */
  int64_t n_terms;               /* affine operator terms */
  const double* term_values;     /* n_terms x nnz: value vector of each term */
  const double* term_coef;       /* nt x n_mu x n_terms: theta_q(mu, t) of step s at [s][mu][q] */
  const double* tril_values;     /* nnz or NULL: T of the state-dependent term diag(u*) T (trilinear, rom.py:931-952) */
  int64_t n_rhs;
  const double* rhs_terms;       /* n_rhs x N: source vectors (lifting / forcing) */
  const double* rhs_coef;        /* nt x n_mu x n_rhs */
} rt_sweep_desc;
/* K_N = bdf M_N + dt V^T(sum_q theta_q A_q + diag(u*) T)V,  b_N = M_N(2u^n - u^{n-1}/2) + dt V^T f,
 * u^{n+1} = K_N^-1 b_N, for all n_mu parameter points per step, nt steps, zero initial condition
 * (rom.py:451-453).  uN_out: n_mu x nt x r (device).  Everything stays on the ctx stream. */
int rt_rom_bdf_sweep(rt_ctx* ctx, const rt_sweep_desc* desc, double* uN_out);

/* ---- hyper-reduced online sweep (the (M)DEIM path of RomConstructor*.solve: interpolate(which=ROM),
 *      deim.py:416-452,477-493, mdeim.py:230-261; assemble_system rom.py:877-929) on the device ------------- */
typedef struct {
  int64_t r, n_mu, nt;           /* reduced size (<= 128), parameter points, time steps */
  double dt;
  int bdf2;                      /* as rt_sweep_desc */
  int64_t m_mass, m_lin, m_nl, m_rhs; /* interpolation coefficients: mass operator, the other (mu,t)-dependent
                                    operators together, the state-dependent operator, the source vectors */
  const double* Z;               /* (m_mass + m_lin + m_nl) x (r*r): row e = the r x r matrix that local entry e
                                    multiplies = column e of basis_rom PT_U^-1 (mdeim.py:153-192 with the theta
                                    solve of deim.py:491-492 folded in); blocks in the order mass | lin | nl */
  const double* Zf;              /* m_rhs x r: the same for the DEIM source vectors (deim.py:495-515) */
  const double* F_mass;          /* nt x n_mu x m_mass: the operator's entries at its interpolation entries,
                                    assemble(mu, t, entries=dofs) (deim.py:429-433), for every step and mu */
  const double* F_lin;           /* nt x n_mu x m_lin */
  const double* F_rhs;           /* nt x n_mu x m_rhs */
  const double* W;               /* m_nl x r: local entries of the state-dependent operator are S (W u_N* + C)
                                    (N-MDEIM, nonlinear.py:247-283: the trilinear form is linear in u_h = V u_N) */
  const double* C_nl;            /* nt x n_mu x m_nl or NULL */
  const double* S_nl;            /* nt x n_mu or NULL (= 1) */
} rt_hsweep_desc;
/* M_N = sum_e F_mass[e] Z_e;  K_N = bdf M_N + dt (sum_e F_lin[e] Z_e + sum_e S (W u* + C)_e Z_e);
 * b_N = M_N (2u^n - u^{n-1}/2) + dt Zf^T F_rhs;  u* = 2u^n - u^{n-1} (BDF2) or u^n;  zero initial condition.
 * No quantity of size N_h is touched.  uN_out: n_mu x nt x r (device). */
int rt_hrom_bdf_sweep(rt_ctx* ctx, const rt_hsweep_desc* desc, double* uN_out);
/* How the reduced systems of the most recent sweep on this ctx were solved (the reference calls GMRES once per step,
 * rom.py:492; here K_N^-1 is tracked from step to step): stats4 = { Newton-Schulz iterations, systems restarted from
 * K^T/(|K|_1 |K|_inf), systems handed to the pivoted LU, systems solved }.  Synchronises the ctx stream. */
int rt_last_sweep_stats(rt_ctx* ctx, int64_t* stats4);

/* ---- closed-form local assembly of the 1-D P1 operators at (M)DEIM entries (the step before the path) ----------
 * What ``assemble(mu, t, entries=dofs[, u_n])`` returns for the reference's 1-D problems (fom/base.py:523-599 per-entry
 * assembly of the forms in fom/nonlinear.py:374-494; closed forms as in testing/mock.py:30-85), for n_states states at
 * once: out[s][e] = operator(kind; h[s], coef[s], nodal function of state s) at entry (rows[e], cols[e]) of the
 * (nx + 1) x (nx + 1) matrix, or at vector entry rows[e] for RT_P1_LOAD (cols = NULL).  h[s] = L(mu, t) / nx is the
 * cell size of state s, coef[s] a scalar factor (NULL = 1; the diffusivity alpha(mu, t) for the stiffness).  The
 * nodal function (w of the trilinear form int w u' v, f of the load int f v): state_mode 1 = state is n_states x (nx+1)
 * nodal values, 2 = state is n_states amplitudes of the ramp amp * node / nx (the piston lifting g = amp x / L), 0 = none.
 * First / last dof are Dirichlet rows (identity rows, zero load).  Feeds the F tables of rt_hrom_bdf_sweep. */
#define RT_P1_MASS 0
#define RT_P1_STIFFNESS 1
#define RT_P1_CONVECTION 2
#define RT_P1_TRILINEAR 3
#define RT_P1_LOAD 4
/* RT_P1_LOAD_P2: int f v with f the P2 interpolant of the data on every cell - what FEniCS integrates for an
 * Expression(..., degree=2) (fom/heat.py:119 forcing; fom/base.py:452-495 lifting data), exact for quadratic f:
 * out = coef h/3 (f_{i-1/2} + f_i + f_{i+1/2}).  state_mode 1 = n_states x (2 nx + 1) values (vertex k at 2k, midpoint
 * of cell k at 2k+1), state_mode 3 = n_states x 3 coefficients of a0 + a1 x + a2 x^2 in the physical coordinate.
 * Reproduces the reference's known-answer tables expected_mat_fh / expected_mat_fgh_time (tests/test_mpf1.py:288-302):
 * forcing problems/mfp1.py:38-39; lifting vector = -h dg_dt(x_i) (fom/heat.py:131-169) = this kind with coef = -1. */
#define RT_P1_LOAD_P2 5
int rt_p1_local_assembly(rt_ctx* ctx, int kind, int64_t nx, const int64_t* rows, const int64_t* cols, int64_t m,
                         int64_t n_states, const double* h, const double* coef, int state_mode, const double* state,
                         double* out);

/* ---- small symmetric eigenproblem of the Gram matrix, on the device (3 <= n <= 1024) ---------- */
/* Householder tridiagonalisation (32 cooperating workgroups, 128 for n > 512; matrix resident in LDS) + Sturm
 * multisection:
 * lam (n, device) = all eigenvalues of the symmetric G (n x n row-major, not modified), DESCENDING.
 * status (device int, may be NULL): 0, or 1 if the inter-workgroup hand-off timed out (results
 * invalid).  Replaces the eigenvalue half of LAPACK's work inside scipy.linalg.svd (pod.py:38). */
int rt_sym_eig_values(rt_ctx* ctx, const double* G, int64_t n, double* lam, int* status);
/* The same, but only the eigenvalues with descending index in [first, first + count) are searched and written
 * (lam[first .. first+count)); the tridiagonalisation is complete either way.  For a row-sharded POD every rank
 * holds the same G after the all-reduce: the ranks split the multisection (and the eigenvectors, by passing
 * lam + first and their share of k to rt_sym_eig_vectors) and all-gather the pieces. */
int rt_sym_eig_values_part(rt_ctx* ctx, const double* G, int64_t n, int64_t first, int64_t count, double* lam,
                           int* status);
/* Eigenvectors of the k LARGEST eigenvalues by inverse iteration on the tridiagonal form and
 * back-transformation; must directly follow rt_sym_eig_values on the same ctx (it reuses the
 * reflectors kept in the ctx's workspace).  W: n x k row-major, column t pairs with lam[t]. */
int rt_sym_eig_vectors(rt_ctx* ctx, int64_t n, int64_t k, const double* lam, double* W);

/* ---- host-side small dense step --------------------------------------------------------- */
/* Cyclic two-sided Jacobi eigen-decomposition of a symmetric PSD n x n HOST matrix A (row-major,
 * destroyed): A = W diag(lam) W^T, lam descending, eigenvectors in the columns of W (row-major).
 * Relative stopping rule |a_pq| <= eps sqrt(a_pp a_qq): small eigenvalues of a graded matrix come
 * out to high relative accuracy (what dgesvd delivers on the snapshots themselves, pod.py:38).
 * HOST pointers; no ctx; used on the second-pass Gram matrix of rt_gram. */
int rt_host_jacobi_eigh(double* A, int64_t n, double* W, double* lam, int max_sweeps, int* sweeps_done);

/* ---- measurement helpers (not part of the reference surface) ------------------------------ */
/* Runs `iters` back-to-back v_mfma_f64_16x16x4_f64 per wave on a full-chip grid and returns
 * the achieved TFLOP/s in *tflops (synchronises). */
int rt_bench_mfma_f64(rt_ctx* ctx, int iters, double* tflops);
/* Device-to-device streaming copy of `bytes` bytes; returns GB/s (read+write) (synchronises). */
int rt_bench_copy(rt_ctx* ctx, void* dst, const void* src, int64_t bytes, int reps, double* gbps);

#ifdef __cplusplus
}
#endif
#endif /* ROMTIME_HIP_H */
