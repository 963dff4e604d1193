"""BASELINE.json's configurations at their stated sizes, HIP path against the oracle (VERDICT r01 "configs_untested").

C1  1-D moving-mesh heat equation with tests/test_mdeim.py's parameters at nx = 1000, 64 (mu, t) snapshots, r = 10,
    through the class surface.
C4  one pipeline: POD of 5e5 x 200 operator snapshots -> 120 collateral modes (the deflated route), greedy on that
    basis (all 120 indices), MDEIM.project_basis of all 120 modes onto r = 80.
C5  the online sweep over the full 1e4-step horizon: hyper-reduced sweep at r = 80, 32 parameter points against
    oracle.hrom_solve; direct sweep at a reduced N against oracle.rom_solve_nonlinear with an exact dense solver.
    Both report how the device solved its systems (inverse tracking / restart / LU fallback)."""
import json
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import romtime_oracle as oracle

pytestmark = pytest.mark.gpu
EPS = 2.2e-16
C4_N = 100_000      # N_h of config 4 (nnz = 5 N - 6 rows in the snapshot matrix)
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "configs_report.jsonl")


def _report(**kw):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as fp:
            fp.write(json.dumps(kw) + "\n")
    except OSError:
        pass
    print(json.dumps(kw))


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


# ------------------------------------------------------------------------------------------------ C5
def test_c5_hyper_reduced_sweep_full_horizon():
    """rt_hrom_bdf_sweep, nt = 1e4, 32 mu, r = 80, 280 interpolation coefficients, against oracle.hrom_solve (the
    reference's per-step theta solves + dense solve, rom.py:430-555 with deim.py:416-452) for 3 parameter points
    over the WHOLE trajectory: north-star bar 1e-10 rel-L2, also at the last step alone (drift of the tracked
    inverse would show there first)."""
    from romtime_amd._lib import Context
    from romtime_amd.sweep import hrom_bdf_sweep
    from romtime_amd.testing.workloads import c5_hyper_reduced

    nt, n_mu, r = 10_000, 32, 80
    terms, d, V, mus = c5_hyper_reduced(nt=nt, n_mu=n_mu, r=r)
    uN = hrom_bdf_sweep(terms["mass"], terms["lin"], terms["nl"], terms["rhs"], terms["dt"], bdf2=True)
    stats = Context.current().sweep_stats()
    uN = uN.cpu().numpy()
    assert uN.shape == (n_mu, nt, r) and np.all(np.isfinite(uN))
    worst, worst_last = 0.0, 0.0
    for b in (0, 13, 31):
        ref = oracle.hrom_solve(terms["mass"], terms["lin"], terms["nl"], terms["rhs"], b, r, nt, terms["dt"], True)
        whole, last = _rel(uN[b].T, ref), _rel(uN[b, -1], ref[:, -1])
        worst, worst_last = max(worst, whole), max(worst_last, last)
        assert whole <= 1e-10 and last <= 1e-10, (b, whole, last)
        assert np.abs(ref).max() > 1e-3                     # a non-trivial trajectory
    # every system was solved once per step; inverse tracking carried all but the first step of each
    assert stats["solves"] == nt * n_mu
    assert stats["restarts"] <= n_mu and stats["lu_fallbacks"] == 0, stats
    _report(config="C5 hyper-reduced sweep 1e4 steps x 32 mu r=80", rel_l2_whole=worst, rel_l2_last_step=worst_last,
            **stats, newton_iterations_per_solve=stats["newton_iterations"] / stats["solves"])


def test_c5_direct_sweep_full_horizon_reduced_N():
    """rt_rom_bdf_sweep for nt = 1e4 BDF2 steps at N = 3000, r = 24, against the oracle's restatement of the
    reference loop (5 separate projections per step, rom.py:877-929) with an exact dense solver, whole trajectory."""
    from romtime_amd._lib import Context
    from romtime_amd.sweep import rom_bdf_sweep
    from romtime_amd.testing.workloads import c5_direct

    nt, N, r = 10_000, 3000, 24
    fom, V, _, _ = c5_direct(N=N, r=r, n_mu=1, nt=nt, dt=1e-3, seed=3)
    mus = [dict(alpha=0.5 + 0.2 * i, beta=1.0 - 0.1 * i, delta=0.3 + 0.05 * i, omega=7.0 + i) for i in range(3)]
    d = fom.descriptor(mus)
    uN = rom_bdf_sweep(V, d["indptr"], d["indices"], d["mass"], d["terms"], d["term_coef"], d["tril"], d["rhs_terms"],
                       d["rhs_coef"], d["dt"], bdf2=True)
    stats = Context.current().sweep_stats()
    uN = uN.cpu().numpy()
    worst = 0.0
    for i, mu in enumerate(mus):
        ref, _ = oracle.rom_solve_nonlinear(fom, V, mu, solver=np.linalg.solve)
        whole, last = _rel(uN[i].T, ref), _rel(uN[i, -1], ref[:, -1])
        worst = max(worst, whole, last)
        assert whole <= 1e-10 and last <= 1e-10, (i, whole, last)
        assert np.abs(ref).max() > 1e-4
    assert stats["solves"] == nt * len(mus) and stats["lu_fallbacks"] == 0, stats
    _report(config="C5 direct sweep 1e4 steps N=3000 r=24", rel_l2=worst, **stats)


def test_c5_direct_sweep_trajectory_at_full_size():
    """rt_rom_bdf_sweep at config 5's own size - N = 1e5, r = 80, 32 parameter points - for 50 BDF2 steps against the
    oracle's restatement of the reference loop (rom.py:430-555, 877-929: five csr.dot + matmul projections per step)
    with an EXACT dense solver, two parameter points, whole reduced trajectory at the north star's 1e-10.  (The
    residual test below validates the solves; this one validates the projections and the loop too.)"""
    from romtime_amd._lib import Context
    from romtime_amd.sweep import rom_bdf_sweep
    from romtime_amd.testing.mock import AffineBurgers
    from romtime_amd.testing.workloads import c5_direct

    nt, n_mu = 50, 32
    fom, V, mus, d = c5_direct(nt=nt, n_mu=n_mu)
    uN = rom_bdf_sweep(V, d["indptr"], d["indices"], d["mass"], d["terms"], d["term_coef"], d["tril"], d["rhs_terms"],
                       d["rhs_coef"], d["dt"], bdf2=True)
    stats = Context.current().sweep_stats()
    uN = uN.cpu().numpy()
    assert uN.shape == (n_mu, nt, 80)
    worst = 0.0
    for b in (0, 31):
        ref, _ = oracle.rom_solve_nonlinear(fom, V, mus[b], solver=np.linalg.solve)
        whole, last = _rel(uN[b].T, ref), _rel(uN[b, -1], ref[:, -1])
        worst = max(worst, whole, last)
        assert whole <= 1e-10 and last <= 1e-10, (b, whole, last)
        assert np.abs(ref).max() > 1e-5
    assert stats["solves"] == nt * n_mu and stats["lu_fallbacks"] == 0, stats
    _report(config="C5 direct sweep 50 steps N=1e5 r=80 32 mu vs exact-solver oracle", rel_l2=worst, **stats)


def test_c5_reduced_residual_inside_reference_acceptance_ball():
    """The reference accepts any u_N with ||K_N u_N - b_N|| <= 1e-10 ||b_N|| (GMRES, rom.py:36,492, info ignored).
    The device's answers at full size (N = 1e5, r = 80, 32 mu) satisfy that at every step: K_N and b_N are rebuilt
    from the public operators for the device's own trajectory and the residual is measured."""
    import torch

    from romtime_amd import ops
    from romtime_amd.sweep import rom_bdf_sweep
    from romtime_amd.testing.workloads import c5_direct

    nt, n_mu = 12, 32
    fom, V, mus, d = c5_direct(nt=nt, n_mu=n_mu)
    N, r = V.shape
    uN = rom_bdf_sweep(V, d["indptr"], d["indices"], d["mass"], d["terms"], d["term_coef"], d["tril"], d["rhs_terms"],
                       d["rhs_coef"], d["dt"], bdf2=True)
    Vd = ops.to_device(V)
    ip, ix = ops.to_device_index(d["indptr"]), ops.to_device_index(d["indices"])
    mass, terms, tril = (ops.to_device(d[k]) for k in ("mass", "terms", "tril"))
    rr = torch.repeat_interleave(torch.arange(N, device="cuda"), torch.from_numpy(np.diff(d["indptr"])).cuda())
    MN = ops.project_csr(ip, ix, mass, Vd)
    fN = ops.gemm_tn(Vd, ops.to_device(d["rhs_terms"]).T.contiguous())
    zero = torch.zeros((n_mu, r), dtype=torch.float64, device="cuda")
    worst = 0.0
    for step in range(nt):
        un = uN[:, step - 1] if step >= 1 else zero
        um = uN[:, step - 2] if step >= 2 else zero
        ustar = (2.0 * un - um) @ Vd.T                                           # n_mu x N
        bdf = 1.5 if step > 0 else 1.0
        kv = bdf * mass[None, :] + d["dt"] * (ops.to_device(d["term_coef"][step]) @ terms + ustar[:, rr] * tril[None, :])
        KN = ops.project_csr_batched(ip, ix, kv.T, Vd)
        bN = (2.0 * un - 0.5 * um) @ MN.T + d["dt"] * (ops.to_device(d["rhs_coef"][step]) @ fN.T)
        res = torch.einsum("bij,bj->bi", KN, uN[:, step]) - bN
        worst = max(worst, float((res.norm(dim=1) / bN.norm(dim=1)).max().item()))
    assert worst <= 1e-10, worst
    _report(config="C5 direct sweep: reduced residual ||K_N u - b_N|| / ||b_N||, N=1e5 r=80 32 mu", worst=worst)


# ------------------------------------------------------------------------------------------------ C4
def test_c4_pipeline_full_size():
    """POD (deflated levels) -> greedy -> project_basis at config 4's sizes, each stage against the oracle."""
    import torch
    from scipy.sparse import csr_matrix

    from romtime_amd import MatrixDiscreteEmpiricalInterpolation, ops, pod
    from romtime_amd.testing.workloads import c4_snapshots

    N, m, r = C4_N, 120, 80
    A, S = c4_snapshots(N=N)
    nnz = A.nnz
    out = pod.pod_device(S, num=m, normalize=False)
    assert out["passes"] == "deflate" and out["r"] == m
    Q = out["Q"].cpu().numpy()
    Sh = S.cpu().numpy()
    del S
    torch.cuda.empty_cache()
    Qo, so, eo = oracle.orth(Sh, num=m, normalize=False)                # dgesvd of 5e5 x 200
    s = out["s"]
    # singular values: the bar of tests/test_surface.py (a deflated level resolves sigma_i to eps sigma_level)
    assert np.all(np.abs(s - so) <= 2e-13 * so[0] + 8 * EPS * so[0] ** 2 / np.maximum(so, 1e-300))
    assert_allclose(out["energy"], eo, rtol=1e-10)
    assert np.abs(Q.T @ Q - np.eye(m)).max() < 1e-9
    # the 8 signal modes individually (well separated), the 112 noise-floor modes (one cluster, gaps ~1e-4) as a subspace
    for i in range(8):
        gaps = np.abs(so - so[i]) / so[i]
        gaps[i] = np.inf
        tol = 1e-10 + 200 * EPS * (so[0] / so[i]) / min(gaps.min(), 1.0)
        err = min(np.linalg.norm(Q[:, i] - Qo[:, i]), np.linalg.norm(Q[:, i] + Qo[:, i]))
        assert err <= tol, (i, err, tol)
    gap = (so[m - 1] - so[m]) / so[0]
    sub = np.linalg.norm(Q @ (Q.T @ Qo) - Qo, 2)
    assert sub <= 1e-10 + 50 * EPS / gap, (sub, gap)
    # greedy on the device's basis, all 120 indices, against the oracle's gather form of deim.py:517-561
    rows = np.repeat(np.arange(N), np.diff(A.indptr))
    md = MatrixDiscreteEmpiricalInterpolation(assemble=None, name="c4")
    md.rows, md.cols = list(rows), list(A.indices)
    md.load_fom_basis(basis=Q)
    dofs_o, PT_U_o, margin = oracle.deim_greedy(Q)
    mine = np.array([A.indptr[i] + int(np.nonzero(A.indices[A.indptr[i]:A.indptr[i + 1]] == j)[0][0]) for (i, j) in md.dofs])
    differ = np.nonzero(mine != dofs_o)[0]
    assert differ.size == 0 or margin[differ[0]] < 1e-9, (differ[:4], margin[differ[:4]])   # bit-exact unless a near-tie
    if differ.size == 0:
        np.testing.assert_array_equal(md.PT_U, PT_U_o)
    # project_basis of all 120 modes; oracle on 8 sampled ones (mdeim.py:153-192)
    V, _ = np.linalg.qr(np.random.RandomState(8).standard_normal((N, r)))
    md.project_basis(V)
    assert md.basis_rom.shape == (r * r, m) and md.N_V == r
    worst = 0.0
    for i in (0, 1, 7, 8, 40, 77, 118, 119):
        ref = oracle.project_csr(csr_matrix((Q[:, i], A.indices, A.indptr), shape=(N, N)), V).flatten()
        worst = max(worst, np.abs(md.basis_rom[:, i] - ref).max() / np.abs(ref).max())
    assert worst <= 1e-12, worst
    _report(config="C4 pipeline 5e5x200 -> 120 modes, greedy, project_basis r=80", sigma_max_rel_err=float(np.abs(s - so).max() / so[0]),
            subspace_dist=float(sub), greedy_indices_equal=bool(differ.size == 0), min_margin=float(margin.min()),
            project_rel_err=float(worst), levels=int(pod.stage_timings().get("levels", 0)) if pod.LAST_TIMINGS else None)


# ------------------------------------------------------------------------------------------------ C1
def _c1_grid():
    from scipy.stats.distributions import uniform

    return {"delta": uniform(0.01, 1.99), "beta": uniform(1.0, 9.0), "alpha_0": uniform(0.01, 1.99)}   # test_mdeim.py:43-47


def test_c1_moving_mesh_heat_at_stated_size():
    """Config 1 at its size: nx = 1000 (N_h = 1001), 8 mu x 8 times = 64 snapshots, r = 10.  POD of the solution
    snapshots against dgesvd; MDEIM of the three operators on the moving mesh (the reference's acceptance test,
    tests/test_mdeim.py:153-228); the projected reductors against the oracle's project_basis and the interpolated
    reduced operator against V^T A V of the assembled one."""
    from sklearn.model_selection import ParameterSampler

    from romtime_amd import MatrixDiscreteEmpiricalInterpolation, orth
    from romtime_amd.conventions import Stage
    from romtime_amd.testing.mock import MockSolver

    nx, r = 1000, 10
    Lt = lambda t, **mu: 1.0 + 0.1 * mu["delta"] * t
    solver = MockSolver(domain={"L0": 1.0, "nx": nx, "T": 5.0, "nt": 100}, Lt=Lt)
    solver.setup()
    ts = np.linspace(0.0, 5.0, 8)
    mus = list(ParameterSampler(_c1_grid(), n_iter=8, random_state=np.random.RandomState(0)))
    # the manufactured solution of the reference's test problem (test_mdeim.py:24-36) on the moved mesh
    snaps = np.array([(1.0 - np.exp(-mu["beta"] * t)) * (1.0 + mu["delta"] ** 2 * solver.x_at(mu, t) ** 2)
                      for mu in mus for t in ts[1:]] + [np.sin((k + 1) * np.pi * np.linspace(0, 1, nx + 1)) * 1e-3
                                                        for k in range(8)]).T
    assert snaps.shape == (nx + 1, 64)
    V, s, energy = orth(snaps, num=r)
    Vo, so, eo = oracle.orth(snaps, num=r)
    assert V.shape == (nx + 1, r)
    assert np.all(np.abs(s - so) <= 2e-13 * so[0] + 8 * EPS * so[0] ** 2 / np.maximum(so, 1e-300))
    assert_allclose(energy, eo, rtol=1e-10)
    for i in range(r):
        gaps = np.abs(so - so[i]) / so[i]
        gaps[i] = np.inf
        tol = 1e-10 + 200 * EPS * (so[0] / so[i]) / min(gaps.min(), 1.0)
        assert min(np.linalg.norm(V[:, i] - Vo[:, i]), np.linalg.norm(V[:, i] + Vo[:, i])) <= tol, i
    report = {}
    for name, assemble in (("stiffness", solver.assemble_stiffness), ("mass", solver.assemble_mass),
                           ("convection", solver.assemble_convection)):
        md = MatrixDiscreteEmpiricalInterpolation(name=name, assemble=assemble, grid=_c1_grid(),
                                                  tree_walk_params={"ts": ts, "num_snapshots": 8})
        md.setup(rnd=np.random.RandomState(0))
        md.run()
        for mu in (md.mu_space[Stage.OFFLINE][0],
                   list(ParameterSampler(_c1_grid(), n_iter=5, random_state=np.random.RandomState(19219)))[0]):
            expected = oracle.eliminate_zeros(assemble(mu=mu, t=1.0)).data
            assert_allclose(md.interpolate(mu=mu, t=1.0, which=md.FOM).data, expected, rtol=1e-7, atol=1e-12)
        md.project_basis(V)
        ref = oracle.mdeim_project_basis(md.basis_fom, md.rows, md.cols, V)
        assert_allclose(md.basis_rom, ref, rtol=0, atol=1e-12 * max(1.0, np.abs(ref).max()))
        mu = mus[3]
        AN = md.interpolate(mu=mu, t=2.5, which=md.ROM)
        A = oracle.eliminate_zeros(assemble(mu=mu, t=2.5).copy())
        A.data[0] = 0.0            # MDEIM drops the Dirichlet entry (0, 0) from its snapshots (deim.py:388-389); only
        exact = oracle.project_csr(A, V)   # the FOM-form interpolant puts it back (deim.py:449-450)
        assert AN.shape == (r, r)
        assert_allclose(AN, exact, rtol=0, atol=1e-7 * np.abs(exact).max())
        report[name] = dict(modes=int(md.N), rom_err=float(np.abs(AN - exact).max() / np.abs(exact).max()))
    _report(config="C1 nx=1000, 64 snapshots, r=10", sigma_rel_err=float(np.abs(s - so).max() / so[0]), **report)
