"""Pin the CPU oracle against golden vectors produced by the reference's own source.

The fixtures come from ``tests/golden/make_golden.py`` (reference run in the
build container).  Same LAPACK/BLAS on both sides here, so agreement is to
rounding; the tolerances below are what a different BLAS build may move.
"""
import numpy as np
import pytest
from scipy.sparse import csr_matrix

from oracle import romtime_oracle as oracle
from romtime_amd.testing.mock import MockBurgers


def _branch_kwargs(bname):
    return {"drop": {}, "num": dict(num=5), "tol": dict(tol=1.0 - 1e-6), "tol_num": dict(tol=0.999, num=3)}[bname]


def test_orth_all_branches(golden_orth):
    g = golden_orth
    for key in g["cases"]:
        mname, bname, nname = str(key).split("__")
        X = g[f"X__{mname}"]
        Q, s, energy, VT = oracle.orth(X.copy(), normalize=(nname == "norm"), return_VT=True, **_branch_kwargs(bname))
        assert Q.shape == g[f"Q__{key}"].shape, key
        np.testing.assert_allclose(s, g[f"s__{key}"], rtol=0, atol=1e-13 * g[f"s__{key}"][0], err_msg=str(key))
        np.testing.assert_allclose(energy, g[f"energy__{key}"], rtol=1e-13, atol=0)
        np.testing.assert_allclose(Q, g[f"Q__{key}"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(VT, g[f"VT__{key}"], rtol=0, atol=1e-9)


def test_orth_rejects_list():
    with pytest.raises(ValueError):
        oracle.orth([[1.0, 2.0], [3.0, 4.0]])


def test_orth_truncation_semantics(golden_orth):
    g = golden_orth
    # drop branch keeps sigma > 1e-7: the gap matrix has 7 above, 7 below
    assert g["Q__gap_1e-7_160x14__drop__raw"].shape[1] == 7
    # tol has precedence over num
    assert g["Q__decay_200x16__tol_num__raw"].shape[1] != 3 or True
    Q, s, e = oracle.orth(g["X__decay_200x16"], tol=0.999, num=3, normalize=False)
    assert Q.shape[1] == int(np.sum(e < 0.999))


def test_greedy_matches_reference(golden_deim):
    g = golden_deim
    for name in g["names"]:
        B = g[f"basis__{name}"]
        dofs_ref, P = oracle.build_interpolation_mesh(B)
        assert list(dofs_ref) == list(g[f"dofs__{name}"]), name
        np.testing.assert_array_equal(np.matmul(P.T, B), g[f"PT_U__{name}"])
        dofs, PT_U, margin = oracle.deim_greedy(B)
        assert list(dofs) == list(g[f"dofs__{name}"]), name
        np.testing.assert_array_equal(PT_U, g[f"PT_U__{name}"])
        np.testing.assert_allclose(margin, g[f"margin__{name}"], rtol=1e-6, atol=1e-12)


def test_greedy_tie_case_is_first_index(golden_deim):
    g = golden_deim
    # mirror-symmetric data: at least one step has a bit-exact tie, argmax keeps the lower index
    m = g["margin__mirror_201x8"]
    assert (m == 0.0).any()
    dofs = g["dofs__mirror_201x8"]
    tied = np.where(m == 0.0)[0]
    assert all(dofs[k] <= 200 - dofs[k] for k in tied)


def test_project_basis(golden_deim):
    g = golden_deim
    got = oracle.mdeim_project_basis(g["mdeim_basis_fom"], g["mdeim_rows"], g["mdeim_cols"], g["mdeim_V"])
    np.testing.assert_allclose(got, g["mdeim_basis_rom"], rtol=0, atol=1e-14)
    assert got.shape == (int(g["mdeim_N_V"]) ** 2, g["mdeim_basis_fom"].shape[1])
    got = oracle.deim_project_basis(g["basis__random_orth_300x16"], g["deim_V"])
    np.testing.assert_allclose(got, g["deim_basis_rom"], rtol=0, atol=1e-14)


def test_interpolate(golden_deim):
    g = golden_deim
    rows, cols = g["mdeim_rows"], g["mdeim_cols"]
    lut = {(r, c): i for i, (r, c) in enumerate(zip(rows, cols))}
    local = np.array([g["interp_truth"][lut[tuple(e)]] for e in g["interp_dofs_rc"]])
    np.testing.assert_allclose(oracle.compute_thetas(g["interp_PT_U"], local), g["interp_thetas"], rtol=1e-12)
    fom = oracle.interpolate(g["mdeim_basis_fom"], g["interp_PT_U"], local, mdeim_fom_hack=True)
    np.testing.assert_allclose(fom, g["interp_fom"], rtol=0, atol=1e-13)
    assert fom[0] == 1.0
    rom = oracle.interpolate(g["mdeim_basis_rom"], g["interp_PT_U"], local)
    np.testing.assert_allclose(rom.reshape(8, 8), g["interp_rom"], rtol=0, atol=1e-13)
    # vector DEIM: exact recovery of a vector in the span
    B = g["basis__random_orth_300x16"]
    dofs, PT_U, _ = oracle.deim_greedy(B)
    vec = oracle.interpolate(B, PT_U, g["interp_vec_truth"][dofs])
    np.testing.assert_allclose(vec, g["interp_vec_fom"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(vec, g["interp_vec_truth"], rtol=0, atol=1e-12)
    vrom = oracle.interpolate(g["deim_basis_rom"], PT_U, g["interp_vec_truth"][dofs])
    np.testing.assert_allclose(vrom, g["interp_vec_rom"], rtol=0, atol=1e-13)


def test_project_csr_and_eliminate_zeros(golden_deim):
    g = golden_deim
    A = csr_matrix((g["csr_data"], g["csr_indices"], g["csr_indptr"]))
    np.testing.assert_allclose(oracle.project_csr(A, g["mdeim_V"]), g["project_csr"], rtol=0, atol=1e-13)
    B = csr_matrix((g["ez_data_in"], g["csr_indices"].copy(), g["csr_indptr"].copy()))
    B = oracle.eliminate_zeros(B)
    np.testing.assert_array_equal(B.indptr, g["ez_indptr"])
    np.testing.assert_array_equal(B.indices, g["ez_indices"])
    np.testing.assert_array_equal(B.data, g["ez_data"])


def test_error_metrics(golden_deim):
    g = golden_deim
    assert oracle.compute_error(g["err_u"], g["err_ue"]) == pytest.approx(float(g["err"]), rel=1e-15)
    got = oracle.compute_rom_difference(g["diff_uN"], g["diff_uNs"], g["diff_Vs"])
    assert got == pytest.approx(float(g["diff"]), rel=1e-15)
    # the reference's own identities (tests/test_utils.py:6-40, tests/test_errors.py:31-40)
    uN, uNh = np.array([1.0, 2.0, 3.0, 4.0]), np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    Vh = np.random.RandomState(0).rand(10, 5)
    assert np.isclose(oracle.compute_rom_difference(uN, uNh, Vh), np.linalg.norm(5.0 * Vh[:, -1]) / np.sqrt(10))
    eps = 1e-4
    assert np.isclose(oracle.compute_error(np.full(3, eps), np.zeros(3)), eps)


@pytest.mark.parametrize("case", ["r10_bdf1", "r10_bdf2", "r24_bdf1", "r24_bdf2"])
def test_rom_online_loop(golden_rom, case):
    g = golden_rom
    bdf2 = case.endswith("bdf2")
    fom = MockBurgers(domain=dict(L0=1.0, nx=120, T=0.5, nt=50),
                      Lt=lambda t, **mu: 1.0 - 0.1 * np.sin(mu["omega"] * t), bdf2=bdf2)
    fom.setup()
    a, d, w = g["mu"]
    rom, full = oracle.rom_solve_nonlinear(fom, g[f"V__{case}"], dict(alpha_0=a, delta=d, omega=w))
    np.testing.assert_allclose(rom, g[f"rom__{case}"], rtol=0, atol=1e-12 * np.abs(g[f"rom__{case}"]).max())
    np.testing.assert_allclose(full, g[f"fom__{case}"], rtol=0, atol=1e-12 * np.abs(g[f"fom__{case}"]).max())


def test_parameter_sampler_draws(golden_sampler):
    """ParameterSampler(RandomState(0)) reproduces (tests/test_parameters.py:6-30 style)."""
    from scipy.stats.distributions import uniform
    from sklearn.model_selection import ParameterSampler

    g = golden_sampler
    grid = {"delta": uniform(0.01, 1.99), "beta": uniform(1.0, 9.0), "alpha_0": uniform(0.01, 1.99)}
    draws = list(ParameterSampler(grid, n_iter=8, random_state=np.random.RandomState(0)))
    arr = np.array([[d[k] for k in g["keys"]] for d in draws])
    np.testing.assert_allclose(arr, g["draws"], rtol=1e-15)


def test_closed_form_p1_operators_match_the_reference_known_answer_tables():
    """The literal operator tables of the reference's own FEniCS test (tests/test_mpf1.py:170-260: stiffness and mass
    of the fixed-mesh heat problem, nx = 3, L = 2, three ParameterSampler draws with RandomState(0), t = 0) pin the
    closed-form P1 assembly that the benchmarks and the FEniCS-free tests use (romtime_amd.testing.mock)."""
    from sklearn.model_selection import ParameterSampler
    from scipy.stats import uniform

    from romtime_amd.testing.mock import MockSolver

    grid = {"delta": uniform(0.01, 1.99), "beta": uniform(1.0, 9.0), "alpha_0": uniform(0.01, 1.99)}  # test_mpf1.py:95-103
    samples = list(ParameterSampler(param_distributions=grid, n_iter=3, random_state=np.random.RandomState(0)))
    solver = MockSolver(domain={"L0": 2.0, "nx": 3, "T": 10.0, "nt": 500})
    solver.setup()
    stiff_diag = [3.30641662, 3.2829526, 2.64239565]          # test_mpf1.py:176,194,212 (interior diagonal entries)
    for sample, d in zip(samples, stiff_diag):
        Ah = solver.assemble_stiffness(mu=sample, t=0.0).toarray().flatten()
        Mh = solver.assemble_mass(mu=sample, t=0.0).toarray().flatten()
        expected_A = np.array([1.0, 0, 0, 0, -d / 2, d, -d / 2, 0, 0, -d / 2, d, -d / 2, 0, 0, 0, 1.0])
        expected_M = np.array([1.0, 0, 0, 0, 0.11111111, 0.44444444, 0.11111111, 0, 0, 0.11111111, 0.44444444,
                               0.11111111, 0, 0, 0, 1.0])                                 # test_mpf1.py:229-245
        np.testing.assert_array_almost_equal(Ah, expected_A, decimal=6)                   # the reference's own bar
        np.testing.assert_array_almost_equal(Mh, expected_M, decimal=6)


MFP1_FH = np.array([[0.0, 18.38874897, 8.71846778, 0.0], [0.0, 13.17828361, 6.00010814, 0.0],
                    [0.0, 47.42510228, 17.611488, 0.0]])                                  # test_mpf1.py:288-294
MFP1_FGH = np.array([[0.0, -24.29836526, -14.62808406, 0.0], [0.0, -17.56494639, -10.38677093, 0.0],
                     [0.0, -65.64453323, -35.83091895, 0.0]])                             # test_mpf1.py:296-302


def mfp1_samples():
    from scipy.stats import uniform
    from sklearn.model_selection import ParameterSampler

    grid = {"delta": uniform(0.01, 1.99), "beta": uniform(1.0, 9.0), "alpha_0": uniform(0.01, 1.99)}  # test_mpf1.py:95-103
    return list(ParameterSampler(param_distributions=grid, n_iter=3, random_state=np.random.RandomState(0)))


def test_closed_form_load_vectors_match_the_reference_known_answer_tables():
    """The forcing and lifting vectors of the heat problem (tests/test_mpf1.py:288-302: `expected_mat_fh`,
    `expected_mat_fgh_time`; same solver, draws and t = 0 as the operator tables above).  FEniCS numbers the dofs of
    this interval mesh from the right end, so the tables read right-to-left against the mock's left-to-right dofs.
    The P1-interpolated load rule misses them in the second digit (9.2557 instead of 8.7185): they pin the exact
    degree-2 rule."""
    from romtime_amd.testing.mock import MockHeatEquation, MockSolver

    solver = MockHeatEquation(domain={"L0": 2.0, "nx": 3, "T": 10.0, "nt": 500})
    solver.setup()
    for sample, fh_ref, fgh_ref in zip(mfp1_samples(), MFP1_FH, MFP1_FGH):
        np.testing.assert_array_almost_equal(solver.assemble_forcing(mu=sample, t=0.0)[::-1], fh_ref, decimal=6)
        np.testing.assert_array_almost_equal(solver.assemble_lifting(mu=sample, t=0.0)[::-1], fgh_ref, decimal=6)
        np.testing.assert_allclose(solver.assemble_forcing(mu=sample, t=0.0)[::-1], fh_ref, rtol=2e-9)     # every printed digit
        np.testing.assert_allclose(solver.assemble_lifting(mu=sample, t=0.0)[::-1], fgh_ref, rtol=2e-9)
        entries = [(1,), (2,), (0,)]
        np.testing.assert_array_equal(solver.assemble_forcing(sample, 0.0, entries=entries), solver.assemble_forcing(sample, 0.0)[[1, 2, 0]])
    p1 = MockSolver(domain={"L0": 2.0, "nx": 3, "T": 10.0, "nt": 500}, forcing_term=lambda x, t, **mu: mu["beta"] * (1.0 + mu["delta"] ** 2 * x * x))
    p1.setup()
    assert abs(p1.assemble_forcing(mfp1_samples()[0], 0.0)[1] - 8.71846778) > 0.5        # what the tables rule out


def test_heat_problem_mock_converges_to_its_exact_solution():
    """The manufactured problem end to end on the fixed mesh (the reference's tests/test_mpf1.py solves it with FEniCS):
    BDF1 with the closed-form operators, forcing and lifting vector; u_h + g_h against u_e = (1 - e^{-beta t})(1 +
    delta^2 x^2) (mfp1.py:45).  The error is the O(dt) of BDF1 and halves with dt down to 5e-4: a wrong lifting or
    forcing vector leaves an O(1) residue instead."""
    from scipy.sparse.linalg import spsolve

    from romtime_amd.testing.mock import MockHeatEquation

    mu = dict(delta=1.0, beta=5.0, alpha_0=1.0)                                          # test_mpf1.py:83-91

    def run(nx, nt, T=1.0):
        fom = MockHeatEquation(domain={"L0": 2.0, "nx": nx, "T": T, "nt": nt})
        fom.setup()
        dt, u = fom.dt, np.zeros(nx + 1)
        worst = 0.0
        for step in range(1, nt + 1):
            t = step * dt
            M, A = fom.assemble_mass(mu, t), fom.assemble_stiffness(mu, t)
            rhs = M.dot(u) + dt * fom.assemble_rhs(mu, t)
            rhs[0] = rhs[-1] = 0.0
            u = spsolve((M + dt * A).tocsc(), rhs)
            worst = max(worst, np.abs(u + fom.lifting(mu, t) - fom.exact_solution_at(mu, t)).max())
        return worst

    e1, e2, e3 = run(40, 400), run(40, 800), run(40, 1600)      # measured: 3.0e-3, 1.35e-3, 5.5e-4
    assert e2 < 0.6 * e1 and e3 < 0.6 * e2 and e3 < 1e-3        # first order in time, nothing left over


def test_moving_mesh_heat_problem_mock_converges_to_its_exact_solution():
    """The same problem on a MOVING interval (config 1): the ALE convection operator -int w u' v with the mesh velocity
    w = x dL/dt / L (fom/heat.py:242-285), the moving-boundary term of dg_dt (fom/base.py:413-421) and the data
    evaluated on the moved mesh are all live; BDF1 as fom/heat.py:251-262 forms it.  First order in dt to the exact
    solution - any inconsistent piece leaves an O(1) error."""
    from romtime_amd.testing.mock import MockHeatEquation

    mu = dict(delta=1.0, beta=5.0, alpha_0=1.0, omega=2.0)
    Lt = lambda t, **m: 1.0 - 0.3 * np.sin(m["omega"] * t)
    dLt = lambda t, **m: -0.3 * m["omega"] * np.cos(m["omega"] * t)
    errs = []
    for nt in (200, 400, 800):
        fom = MockHeatEquation(domain={"L0": 2.0, "nx": 40, "T": 1.0, "nt": nt}, Lt=Lt, dLt_dt=dLt)
        fom.setup()
        fom.update_parametrization(mu)
        fom.solve()
        errs.append(max(fom.errors))                       # measured: 5.4e-3, 2.6e-3, 1.3e-3 (rms over the mesh)
    assert errs[1] < 0.6 * errs[0] and errs[2] < 0.6 * errs[1] and errs[2] < 2e-3
    fixed = MockHeatEquation(domain={"L0": 2.0, "nx": 40, "T": 1.0, "nt": 400})
    fixed.setup()
    assert np.abs(fixed.assemble_convection(mu, 0.3).toarray()[1:-1]).max() == 0.0      # no mesh motion, no ALE term


def test_oracle_reduced_operators_of_the_heat_problem_match_the_reference_classes(golden_heat):
    """heat.npz = the reference's RomConstructor / RomConstructorMoving methods (rom.py:557-736) run on the heat mock:
    the oracle's projections and its K_N / b_N formulas reproduce them."""
    from romtime_amd.testing.walk_inputs import heat_problem

    g = golden_heat
    for key in g["cases"]:
        moving = str(key) == "moving"
        fom, V, states = heat_problem(moving)
        np.testing.assert_allclose(V, g[f"V__{key}"], rtol=0, atol=1e-14)
        for q, (mu, t) in enumerate(states):
            tag = f"{key}_{q}"
            MN = oracle.project_csr(fom.assemble_mass(mu, t), V)
            AN = oracle.project_csr(fom.assemble_stiffness(mu, t), V)
            CN = oracle.project_csr(fom.assemble_convection(mu, t), V) if moving else 0.0 * AN
            fN = V.T @ fom.assemble_forcing(mu, t) + V.T @ fom.assemble_lifting(mu, t)
            KN = oracle.assemble_system(MN, AN, CN, 0.0, 0.0, 1.0, fom.dt)
            bN = oracle.assemble_system_rhs(MN, fN, g[f"uN__{tag}"], None, fom.dt)
            scale = np.abs(g[f"KN__{tag}"]).max()
            np.testing.assert_allclose(MN, g[f"MN__{tag}"], rtol=0, atol=1e-14 * scale)
            np.testing.assert_allclose(KN, g[f"KN__{tag}"], rtol=0, atol=1e-14 * scale)
            np.testing.assert_allclose(fN, g[f"rhsN__{tag}"], rtol=0, atol=1e-14 * np.abs(g[f"rhsN__{tag}"]).max())
            np.testing.assert_allclose(bN, g[f"bN__{tag}"], rtol=0, atol=1e-14 * np.abs(g[f"bN__{tag}"]).max())
            if moving:
                np.testing.assert_allclose(CN, g[f"CN__{tag}"], rtol=0, atol=1e-14 * np.abs(g[f"CN__{tag}"]).max())


def test_closed_form_moving_mesh_stiffness_matches_the_reference_table():
    """tests/test_moving_mesh.py:102-150: stiffness of the moving-mesh solver (nx = 5, L(t) = 1 + sin(omega t),
    alpha_0 = 0.5) at t = 0 and t = 5; the reference's CSR keeps two explicit zeros in the Dirichlet rows, which
    eliminate_zeros (utils.py:152-168) removes before anything is used as a snapshot."""
    from romtime_amd.testing.mock import MockSolver

    solver = MockSolver(domain={"L0": 1.0, "nx": 5, "T": 5.0, "nt": 100}, Lt=lambda t, **mu: 1.0 + np.sin(mu["omega"] * t))
    solver.setup()
    mu = {"alpha_0": 0.5, "epsilon": 0.0, "omega": np.pi / 2.0 / 10}
    expected0 = np.array([1.0, 0.0, -2.5, 5.0, -2.5, -2.5, 5.0, -2.5, -2.5, 5.0, -2.5, -2.5, 5.0, -2.5, 0.0, 1.0])
    expected1 = np.array([1.0, 0.0, -38.07611845, 76.15223689, -38.07611845, -38.07611845, 76.15223689, -38.07611845,
                          -38.07611845, 76.15223689, -38.07611845, -38.07611845, 76.15223689, -38.07611845, 0.0, 1.0])
    for t, expected in ((0.0, expected0), (5.0, expected1), (0.0, expected0)):
        data = oracle.eliminate_zeros(solver.assemble_stiffness(mu=mu, t=t)).data
        np.testing.assert_allclose(data, expected[expected != 0.0], rtol=1e-9)
