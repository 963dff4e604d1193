"""The romtime class surface of romtime_amd against golden vectors from the reference source.

Every check is written once and run twice: on the HIP path (``-m gpu``) and, for the host logic
only, with the device operators stubbed by the oracle's arithmetic (``cpu_ops`` fixture)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose
from scipy.sparse import csr_matrix

from oracle import romtime_oracle as oracle
from romtime_amd.testing.mock import MockBurgers, MockSolver

EPS = 2.2e-16


def _branch_kwargs(bname):
    return {"drop": {}, "num": dict(num=5), "tol": dict(tol=1.0 - 1e-6), "tol_num": dict(tol=0.999, num=3)}[bname]


def pod_column_tolerance(s, i):
    """Stated tolerance for a left singular vector against dgesvd's: both sides carry an error of
    about eps * sigma_1 / (sigma_i * relative gap); 1e-10 is the north-star floor."""
    gaps = np.abs(s - s[i]) / s[i]
    gaps[i] = np.inf
    return 1e-10 + 200 * EPS * (s[0] / s[i]) / min(gaps.min(), 1.0)


def check_orth_golden(g):
    from romtime_amd import orth

    for key in g["cases"]:
        mname, bname, nname = str(key).split("__")
        X = g[f"X__{mname}"]
        Q, s, energy, VT = orth(X.copy(), normalize=(nname == "norm"), return_VT=True, **_branch_kwargs(bname))
        gQ, gs, ge, gVT = g[f"Q__{key}"], g[f"s__{key}"], g[f"energy__{key}"], g[f"VT__{key}"]
        kw = _branch_kwargs(bname)
        knife_edge = "tol" in kw and np.abs(ge - kw["tol"]).min() < 1e-12  # energy == tol to rounding
        assert isinstance(Q, np.ndarray)
        if knife_edge:  # decay_200x16, tol=0.999: energy[2] = 0.999 + 1e-16; either count is "right"
            assert abs(Q.shape[1] - gQ.shape[1]) <= 1
            keep = min(Q.shape[1], gQ.shape[1])
            Q, gQ, VT, gVT = Q[:, :keep], gQ[:, :keep], VT[:keep], gVT[:keep]
        assert Q.shape == gQ.shape, (key, Q.shape, gQ.shape)
        assert s.shape == gs.shape and energy.shape == ge.shape  # ALL singular values are returned
        # singular values: 2e-13 sigma_1 where resolved; a single Gram pass carries an absolute error
        # eps sigma_1^2 in sigma^2, i.e. ~eps sigma_1^2 / sigma_i in sigma_i (tail of the spectrum)
        assert np.all(np.abs(s - gs) <= 2e-13 * gs[0] + 8 * EPS * gs[0] ** 2 / np.maximum(gs, 1e-300)), key
        assert_allclose(energy, ge, rtol=1e-10, atol=0, err_msg=str(key))
        for i in range(Q.shape[1]):
            err = min(np.linalg.norm(Q[:, i] - gQ[:, i]), np.linalg.norm(Q[:, i] + gQ[:, i]))
            assert err <= pod_column_tolerance(gs, i), (key, i, err, gs[i] / gs[0])
            sgn = np.sign(Q[:, i] @ gQ[:, i])
            errv = np.linalg.norm(sgn * VT[i] - gVT[i])
            assert errv <= pod_column_tolerance(gs, i), (key, i, errv)
        # orthonormality of what we return
        if Q.shape[1]:
            assert np.abs(Q.T @ Q - np.eye(Q.shape[1])).max() < 1e-8


def check_orth_semantics():
    from romtime_amd import orth

    with pytest.raises(ValueError):
        orth([[1.0, 2.0], [3.0, 4.0]])
    X = np.random.RandomState(0).standard_normal((50, 6))
    X[:, 2] = 0.0
    with pytest.raises(ValueError):  # zero norm + normalize=True: the reference's svd rejects the NaNs
        orth(X, normalize=True)
    Q, s, e = orth(X, normalize=False)  # rank 5: drop branch removes the null mode
    assert Q.shape == (50, 5) and s.shape == (6,)
    Q1, _, _ = orth(X, normalize=False, passes=1, num=3)
    Q2, _, _ = orth(X, normalize=False, passes=2, num=3)
    for i in range(3):
        assert min(np.linalg.norm(Q1[:, i] - Q2[:, i]), np.linalg.norm(Q1[:, i] + Q2[:, i])) < 1e-12


def check_deim_golden(g):
    from romtime_amd import DiscreteEmpiricalInterpolation, MatrixDiscreteEmpiricalInterpolation

    # greedy + PT_U through the class (load_fom_basis re-derives dofs / PT_U, deim.py:133-164)
    for name in g["names"]:
        d = DiscreteEmpiricalInterpolation(assemble=None, name="golden")
        d.load_fom_basis(basis=g[f"basis__{name}"].copy())
        assert [i for (i,) in d.dofs] == list(g[f"dofs__{name}"]), name
        np.testing.assert_array_equal(d.PT_U, g[f"PT_U__{name}"])
        dofs, P = d.build_interpolation_mesh()
        np.testing.assert_array_equal(np.matmul(P.T, d.basis_fom), g[f"PT_U__{name}"])
        assert np.asarray(P).shape == (d.Nh, d.N) and np.asarray(P).sum() == d.N
    # DEIM.project_basis
    d = DiscreteEmpiricalInterpolation(assemble=None, name="golden")
    d.load_fom_basis(basis=g["basis__random_orth_300x16"].copy())
    d.project_basis(g["deim_V"])
    assert_allclose(d.basis_rom, g["deim_basis_rom"], rtol=0, atol=1e-13)
    tv = g["interp_vec_truth"]
    d.assemble = lambda mu, t, entries=None: np.array([tv[i] for (i,) in entries])
    assert_allclose(d._interpolate(mu={}, t=0.0, which=d.FOM), g["interp_vec_fom"], rtol=0, atol=1e-12)
    assert_allclose(d._interpolate(mu={}, t=0.0, which=d.ROM), g["interp_vec_rom"], rtol=0, atol=1e-12)
    # MDEIM.project_basis + interpolate (FOM form with the Dirichlet hack, ROM form reshaped)
    md = MatrixDiscreteEmpiricalInterpolation(assemble=None, name="golden")
    md.rows, md.cols = list(g["mdeim_rows"]), list(g["mdeim_cols"])
    md.load_fom_basis(basis=g["mdeim_basis_fom"].copy())
    assert [list(rc) for rc in md.dofs] == [list(rc) for rc in g["interp_dofs_rc"]]
    md.project_basis(g["mdeim_V"])
    assert md.N_V == int(g["mdeim_N_V"])
    assert_allclose(md.basis_rom, g["mdeim_basis_rom"], rtol=0, atol=1e-13)
    truth = g["interp_truth"]
    lut = {(r, c): i for i, (r, c) in enumerate(zip(md.rows, md.cols))}
    md.assemble = lambda mu, t, entries=None: np.array([truth[lut[tuple(e)]] for e in entries])
    fom_form = md._interpolate(mu={}, t=0.0, which=md.FOM)
    assert_allclose(fom_form, g["interp_fom"], rtol=0, atol=1e-12)
    assert fom_form[0] == 1.0
    rom_form = md.interpolate(mu={}, t=0.0, which=md.ROM)
    assert rom_form.shape == (8, 8)
    assert_allclose(rom_form, g["interp_rom"], rtol=0, atol=1e-12)
    csr = md.interpolate(mu={}, t=0.0, which=md.FOM)
    assert isinstance(csr, csr_matrix)
    # copies are deep and keep the callback
    c = md.copy()
    c.basis_fom[0, 0] += 1.0
    assert md.basis_fom[0, 0] != c.basis_fom[0, 0] and c.rows == md.rows


def _grid():
    from scipy.stats.distributions import uniform

    return {"delta": uniform(0.01, 1.99), "beta": uniform(1.0, 9.0), "alpha_0": uniform(0.01, 1.99)}


def check_mdeim_end_to_end(operator):
    """tests/test_mdeim.py:153-228 on the closed-form P1 FOM: the MDEIM interpolant reproduces the
    assembled operator on a training parameter and on an unseen one (assert_allclose rtol 1e-7)."""
    from sklearn.model_selection import ParameterSampler

    from romtime_amd import MatrixDiscreteEmpiricalInterpolation
    from romtime_amd.conventions import Stage

    solver = MockSolver(domain={"L0": 1.0, "nx": 100, "T": 5.0, "nt": 100},
                        Lt=(lambda t, **mu: 1.0 + 0.1 * mu["delta"] * t) if operator == "stiffness_ale" else None)
    solver.setup()
    assemble = {"stiffness": solver.assemble_stiffness, "stiffness_ale": solver.assemble_stiffness,
                "mass": solver.assemble_mass, "convection": solver.assemble_convection}[operator]
    ts = np.linspace(0, 5.0, 20)
    mdeim = MatrixDiscreteEmpiricalInterpolation(name=operator, assemble=assemble, grid=_grid(),
                                                 tree_walk_params={"ts": ts, "num_snapshots": 10})
    mdeim.setup(rnd=np.random.RandomState(0))
    mdeim.run()
    assert mdeim.report[Stage.OFFLINE]["basis-shape-final"] == mdeim.N >= 1
    mu = mdeim.mu_space[Stage.OFFLINE][0]
    expected = oracle.eliminate_zeros(assemble(mu=mu, t=1.0)).data
    assert_allclose(mdeim.interpolate(mu=mu, t=1.0, which=mdeim.FOM).data, expected, rtol=1e-7, atol=1e-12)
    mu = list(ParameterSampler(_grid(), n_iter=5, random_state=np.random.RandomState(19219)))[0]
    expected = oracle.eliminate_zeros(assemble(mu=mu, t=1.0)).data
    assert_allclose(mdeim.interpolate(mu=mu, t=1.0, which=mdeim.FOM).data, expected, rtol=1e-7, atol=1e-12)
    mdeim.evaluate(num=3, ts=ts[:4])
    assert all(np.all(e < 1e-8) for e in mdeim.errors_rom.values())


def check_deim_vector_end_to_end():
    """tests/test_deim.py:165-213: DEIM of the forcing vector is exact on train / unseen mu."""
    from romtime_amd import DiscreteEmpiricalInterpolation
    from romtime_amd.conventions import Stage

    forcing = lambda x, t, **mu: (mu["beta"] * np.exp(-mu["beta"] * t) * (1.0 + mu["delta"] ** 2 * x * x)
                                  - 2.0 * mu["delta"] ** 2 * mu["alpha_0"] * (1.0 - np.exp(-mu["beta"] * t)))
    solver = MockSolver(domain={"L0": 1.0, "nx": 200, "T": 5.0, "nt": 100}, forcing_term=forcing)
    solver.setup()
    deim = DiscreteEmpiricalInterpolation(name="forcing", assemble=solver.assemble_forcing, grid=_grid(),
                                          tree_walk_params={"ts": np.linspace(0, 5.0, 15), "num_snapshots": 8})
    deim.setup(rnd=np.random.RandomState(0))
    deim.run()
    mu = deim.mu_space[Stage.OFFLINE][0]
    assert_allclose(deim.interpolate(mu=mu, t=0.7), solver.assemble_forcing(mu, 0.7), rtol=0, atol=1e-12)


def _burgers(bdf2, nx=120, nt=50):
    fom = MockBurgers(domain=dict(L0=1.0, nx=nx, T=0.5, nt=nt),
                      Lt=lambda t, **mu: 1.0 - 0.1 * np.sin(mu["omega"] * t), bdf2=bdf2)
    fom.setup()
    return fom


def check_rom_online_golden(g, case):
    """RomConstructorNonlinear.solve against the reference's own loop (rom.py:430-555,877-929).

    Two bars.  (1) Against the golden trajectory: the reference stops GMRES at a 1e-10 relative
    residual (rom.py:36) and never checks ``info``, which leaves it 1e-8 .. 4e-8 rel-L2 away from
    the exact solution of its own linear systems on these cases (measured: oracle with
    np.linalg.solve vs golden), so the bar here is 2e-7.  (2) Against the oracle loop with an
    exact dense solver -- same assembly, same BDF recurrences, no Krylov tolerance -- the
    north-star bar of 1e-10 rel-L2 applies."""
    from romtime_amd import RomConstructorNonlinear

    fom = _burgers(case.endswith("bdf2"))
    a, d, w = g["mu"]
    mu = dict(alpha_0=a, delta=d, omega=w)
    rom = RomConstructorNonlinear(fom=fom, grid=None, name="golden")
    rom.setup(rnd=0)
    rom.basis = g[f"V__{case}"]
    idx = rom.solve(mu=mu, step="online")
    assert idx == 0
    ref_rom, ref_fom = g[f"rom__{case}"], g[f"fom__{case}"]
    assert rom.solutions.rom.shape == ref_rom.shape and rom.solutions.fom.shape == ref_fom.shape
    assert_allclose(rom.solutions.ts, g[f"ts__{case}"], rtol=1e-14)
    rel = lambda a_, b_: np.linalg.norm(a_ - b_) / np.linalg.norm(b_)
    assert rel(rom.solutions.fom, ref_fom) <= 2e-7 and rel(rom.solutions.rom, ref_rom) <= 2e-7
    exact_rom, exact_fom = oracle.rom_solve_nonlinear(fom, g[f"V__{case}"], mu, solver=np.linalg.solve)
    assert rel(rom.solutions.fom, exact_fom) <= 1e-10, rel(rom.solutions.fom, exact_fom)
    assert rel(rom.solutions.rom, exact_rom) <= 1e-10


def check_rom_offline_and_hyper_reduced():
    """build_reduced_basis (two-level POD) -> truncate -> MDEIM-hyper-reduced online solve."""
    from romtime_amd import MatrixDiscreteEmpiricalInterpolation, RomConstructorNonlinear
    from romtime_amd.conventions import OperatorType, RomParameters, Stage, Treewalk

    fom = _burgers(True, nx=80, nt=30)
    mus = [dict(alpha_0=0.05 + 0.02 * i, delta=0.3, omega=9.0 + i) for i in range(3)]
    srom = RomConstructorNonlinear(fom=fom, grid=None, name="srom")
    srom.setup(rnd=0)
    sols = srom.build_reduced_basis(mu_space=mus, tolerances={RomParameters.TOL_TIME: None, RomParameters.TOL_MU: None})
    assert set(sols) == {0, 1, 2} and srom.basis.shape[0] == fom.Nh
    off = srom.report[Stage.OFFLINE]
    assert off[Treewalk.BASIS_FINAL] == srom.N and len(off[Treewalk.SPECTRUM_TIME]) == 3
    assert np.abs(srom.basis.T @ srom.basis - np.eye(srom.N)).max() < 1e-8
    rom = srom.truncate(2)
    assert rom.N == srom.N - 2
    # ROM reproduces a training trajectory
    rom.solve(mu=mus[1], step=Stage.ONLINE)
    err = np.linalg.norm(rom.solutions.fom - sols[1]) / np.linalg.norm(sols[1])
    assert err < 5e-3, err
    direct = rom.solutions.rom.copy()
    # hyper-reduce the (linear, mu/t dependent) stiffness operator with an MDEIM
    from scipy.stats.distributions import uniform

    grid = dict(alpha_0=uniform(0.04, 0.08), delta=uniform(0.29, 0.02), omega=uniform(8.5, 3.0))
    md = MatrixDiscreteEmpiricalInterpolation(name="stiffness", assemble=fom.assemble_stiffness, grid=grid,
                                              tree_walk_params={"ts": np.linspace(0.01, 0.5, 8), "num_snapshots": 4})
    md.setup(rnd=np.random.RandomState(1))
    md.run()
    rom.add_hyper_reductor(md, OperatorType.STIFFNESS)
    rom.project_reductors()
    assert rom.mdeim_Ah is not md and rom.mdeim_Ah.basis_rom.shape == (rom.N ** 2, md.N)
    rom.solve(mu=mus[1], step=Stage.ONLINE)
    assert_allclose(rom.solutions.rom, direct, rtol=0, atol=1e-7 * np.abs(direct).max())
    with pytest.raises(NotImplementedError):
        rom.add_hyper_reductor(md, "no-such-operator")


def _fully_hyper_reduced_rom():
    """MockBurgers ROM with an (M)DEIM on every operator: mass, stiffness, convection, nonlinear lifting (MDEIM),
    trilinear (N-MDEIM over the reduced basis, nonlinear.py:159-212) and the lifting vector (DEIM)."""
    from scipy.stats.distributions import uniform

    from romtime_amd import (DiscreteEmpiricalInterpolation, MatrixDiscreteEmpiricalInterpolation,
                             MatrixDiscreteEmpiricalInterpolationNonlinear, RomConstructorNonlinear)
    from romtime_amd.conventions import OperatorType, RomParameters

    fom = _burgers(True, nx=60, nt=14)
    mus = [dict(alpha_0=0.05 + 0.02 * i, delta=0.3, omega=9.0 + i) for i in range(3)]
    srom = RomConstructorNonlinear(fom=fom, grid=None, name="srom")
    srom.setup(rnd=0)
    srom.build_reduced_basis(mu_space=mus, tolerances={RomParameters.TOL_TIME: None, RomParameters.TOL_MU: None})
    rom = srom.truncate(max(srom.N - 8, 0)) if srom.N > 8 else srom
    grid = dict(alpha_0=uniform(0.04, 0.08), delta=uniform(0.29, 0.02), omega=uniform(8.5, 3.0))
    tw = {"ts": np.linspace(0.01, 0.5, 8), "num_snapshots": 4}
    rnd = lambda: np.random.RandomState(1)
    ops_m = dict(mass=(fom.assemble_mass, OperatorType.MASS), stiffness=(fom.assemble_stiffness, OperatorType.STIFFNESS),
                 convection=(fom.assemble_convection, OperatorType.CONVECTION),
                 nonlinear_lifting=(fom.assemble_nonlinear_lifting, OperatorType.NONLINEAR_LIFTING))
    for name, (assemble, which) in ops_m.items():
        md = MatrixDiscreteEmpiricalInterpolation(name=name, assemble=assemble, grid=grid, tree_walk_params=tw)
        md.setup(rnd=rnd())
        md.run()
        rom.add_hyper_reductor(md, which)
    nm = MatrixDiscreteEmpiricalInterpolationNonlinear(name="trilinear", assemble=fom.assemble_trilinear, grid=grid,
                                                       tree_walk_params=tw)
    nm.setup(rnd=rnd(), u_n=np.linspace(0.0, 1.0, fom.Nh))
    nm.run(u_n=rom.basis)
    rom.add_hyper_reductor(nm, OperatorType.TRILINEAR)
    dl = DiscreteEmpiricalInterpolation(name="lifting", assemble=fom.assemble_lifting, grid=grid, tree_walk_params=tw)
    dl.setup(rnd=rnd())
    dl.run()
    rom.add_hyper_reductor(dl, OperatorType.LIFTING)
    rom.project_reductors()
    return fom, rom, mus


def check_hyper_reduced_loop_matches_oracle(device_sweep):
    """The fully hyper-reduced online loop: class surface (host-driven, the reference's call sequence) ==
    oracle.hrom_solve on the tables the reductors produce == the device-resident sweep (GPU runs only)."""
    from romtime_amd.conventions import Stage
    from romtime_amd.sweep import hrom_bdf_sweep, hrom_terms_from_rom

    fom, rom, mus = _fully_hyper_reduced_rom()
    assert rom.mdeim_Nh.basis_rom.shape[0] == rom.N ** 2 and rom.deim_fgh.basis_rom.shape[0] == rom.N
    terms = hrom_terms_from_rom(rom, mus)
    r, nt = rom.N, fom.domain["nt"]
    trajectories = []
    for b, mu in enumerate(mus):
        rom.solve(mu=mu, step=Stage.ONLINE)
        host = rom.solutions.rom.copy()
        ref = oracle.hrom_solve(terms["mass"], terms["lin"], terms["nl"], terms["rhs"], b, r, nt, terms["dt"],
                                terms["bdf2"])
        assert host.shape == ref.shape == (r, nt)
        assert np.linalg.norm(host - ref) <= 1e-9 * np.linalg.norm(ref), b
        trajectories.append(ref)
    if device_sweep:
        uN = hrom_bdf_sweep(terms["mass"], terms["lin"], terms["nl"], terms["rhs"], terms["dt"], terms["bdf2"])
        uN = uN.cpu().numpy()
        for b, ref in enumerate(trajectories):
            assert np.linalg.norm(uN[b].T - ref) <= 1e-10 * np.linalg.norm(ref), b
    # sanity only (not a parity bar): the 8-mode, fully hyper-reduced model still follows the full-order solution it
    # was trained on; the level depends on where the tolerance-based truncations of the collateral bases fall
    fom.update_parametrization(mus[1])
    fom.solve()
    rom.solve(mu=mus[1], step=Stage.ONLINE)
    err = np.linalg.norm(rom.solutions.fom - fom.solutions.fom) / np.linalg.norm(fom.solutions.fom)
    assert err < 0.2, err


def check_to_rom_roundtrip():
    from romtime_amd import RomConstructor

    fom = _burgers(True)
    rng = np.random.RandomState(1)
    V, _ = np.linalg.qr(rng.standard_normal((fom.Nh, 9)))
    rom = RomConstructor(fom=fom, grid=None)
    rom.setup(rnd=0)
    rom.basis = V
    mu = dict(alpha_0=0.1, delta=0.2, omega=3.0)
    A = fom.assemble_stiffness(mu, 0.3)
    assert_allclose(rom.to_rom(A), oracle.project_csr(A, V), rtol=0, atol=1e-12)
    f = rng.standard_normal(fom.Nh)
    assert_allclose(rom.to_rom(f), V.T @ f, rtol=0, atol=1e-13)
    uN = rng.standard_normal(9)
    assert_allclose(rom.to_fom_vector(uN), V @ uN, rtol=0, atol=1e-13)
    assert_allclose(rom.to_rom_vector(V @ uN), uN, rtol=0, atol=1e-12)
    assert rom.N == 9 and rom.shape == V.shape


def check_linear_rom_classes_on_the_heat_problem(golden_heat):
    """RomConstructor / RomConstructorMoving (rom.py:34-736) on config 1's problem, the manufactured heat equation on a
    fixed and a moving interval (testing.mock.MockHeatEquation), against the reference's OWN classes run on the same
    FOM (tests/golden/heat.npz: every reduced operator and vector, K_N = M_N + dt (A_N [+ C_N]) and
    b_N = M_N u_N + dt f_N as rom.py:557-573, 714-736 form them).  The reference's `solve` cannot run for these two
    classes at v0 (argument lists of the nonlinear class at the call sites, rom.py:487-488), so the time loop is
    checked end to end instead: reduced basis from three parameters, online solve for a fourth, against the FOM's own
    run and the problem's exact solution."""
    from romtime_amd import RomConstructor, RomConstructorMoving
    from romtime_amd.conventions import RomParameters, Stage
    from romtime_amd.testing.walk_inputs import heat_problem

    g = golden_heat
    for key in g["cases"]:
        moving = str(key) == "moving"
        fom, V, states = heat_problem(moving)
        assert_allclose(V, g[f"V__{key}"], rtol=0, atol=1e-14)
        rom = (RomConstructorMoving if moving else RomConstructor)(fom=fom, grid=None)
        rom.setup(rnd=0)
        rom.basis = g[f"V__{key}"]
        assert int(g[f"n_states__{key}"]) == len(states)
        for q, (mu, t) in enumerate(states):
            tag = f"{key}_{q}"
            scale = np.abs(g[f"KN__{tag}"]).max()
            host = lambda a: a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
            MN, KN = rom.assemble_system(mu, t)        # device tensors here: `solve` keeps the step's r x r algebra in HBM
            assert_allclose(host(MN), g[f"MN__{tag}"], rtol=0, atol=1e-13 * scale)
            assert_allclose(host(KN), g[f"KN__{tag}"], rtol=0, atol=1e-13 * scale)
            assert_allclose(rom.assemble_stiffness(mu, t), g[f"AN__{tag}"], rtol=0, atol=1e-13 * np.abs(g[f"AN__{tag}"]).max())
            for name, fn in (("fN", rom.assemble_forcing), ("fgN", rom.assemble_lifting), ("rhsN", rom.assemble_rhs)):
                ref = g[f"{name}__{tag}"]
                assert_allclose(fn(mu, t), ref, rtol=0, atol=1e-13 * max(np.abs(ref).max(), 1e-300), err_msg=name)
            bN = rom.assemble_system_rhs(mu, t, MN, g[f"uN__{tag}"])          # the call-site order (rom.py:488)
            assert_allclose(host(bN), g[f"bN__{tag}"], rtol=0, atol=1e-13 * np.abs(g[f"bN__{tag}"]).max())
            if moving:
                ref = g[f"CN__{tag}"]
                assert_allclose(rom.assemble_convection(mu, t), ref, rtol=0, atol=1e-13 * np.abs(ref).max())
        # the time loop: offline on three parameters, online on a fourth
        fom2, _, _ = heat_problem(moving, nx=120, nt=60)
        rom = (RomConstructorMoving if moving else RomConstructor)(fom=fom2, grid=None)
        rom.setup(rnd=0)
        train = [dict(delta=0.5 + 0.4 * i, beta=3.0 + 2.0 * i, alpha_0=0.5 + 0.3 * i, omega=1.0 + 0.7 * i) for i in range(3)]
        rom.build_reduced_basis(mu_space=train, tolerances={RomParameters.TOL_TIME: 1.0 - 1e-12, RomParameters.TOL_MU: 1.0 - 1e-12})
        assert 3 <= rom.N <= 40
        mu = dict(delta=0.8, beta=4.5, alpha_0=0.7, omega=1.5)
        fom2.update_parametrization(mu)
        fom2.solve()
        truth = fom2.solutions.fom.copy()
        rom.solve(mu=mu, step=Stage.ONLINE)
        err = np.linalg.norm(rom.solutions.fom - truth) / np.linalg.norm(truth)
        assert err < 5e-4, (key, err, rom.N)                       # the basis reproduces an unseen parameter's FOM run
        exact = np.array([fom2.exact_solution_at(mu, t) for t in fom2.solutions.ts]).T
        assert np.abs(rom.solutions.fom - exact).max() < 10 * max(fom2.errors) + 1e-3


def check_artefact_round_trip(golden_deim, tmp_path, monkeypatch):
    """On-disk artefacts keep the reference's names and payloads (deim.py:77-81,166-173; conventions.py:4-12;
    hrom.py:151-166): a basis dumped by one reductor is adopted by a fresh one under the same name, and the
    ROM basis / parameter-space files use the StorageNames strings."""
    import json
    import pickle

    from romtime_amd.conventions import StorageNames
    from romtime_amd.deim import DiscreteEmpiricalInterpolation
    from romtime_amd.mdeim import MatrixDiscreteEmpiricalInterpolation
    from romtime_amd.utils import dump_pickle, read_pickle

    monkeypatch.chdir(tmp_path)
    assert (StorageNames.ROM, StorageNames.SROM, StorageNames.MU_SPACE, StorageNames.MU_SPACE_DEIM) == (
        "basis_rom.pkl", "basis_srom.pkl", "mu_space.json", "mu_space_deim.json")
    g = golden_deim
    basis = g["basis__random_orth_300x16"].copy()
    d = DiscreteEmpiricalInterpolation(assemble=None, name="RHS  Forcing")
    assert d.basis_pickle_name == "basis_fom_deim_rhs_forcing.pkl"
    md = MatrixDiscreteEmpiricalInterpolation(assemble=None, name="Mass")
    assert md.basis_pickle_name == "basis_fom_mdeim_mass.pkl"
    d.load_fom_basis(basis=basis)
    d.dump_fom_basis()
    with open("basis_fom_deim_rhs_forcing.pkl", "rb") as fp:   # a plain pickled ndarray, as the reference writes
        raw = pickle.load(fp)
    assert isinstance(raw, np.ndarray) and np.array_equal(raw, basis)
    d2 = DiscreteEmpiricalInterpolation(assemble=None, name="rhs forcing")
    d2.load_fom_basis(keep=12)                                   # from disk, truncated (hrom.py:389)
    assert d2.N == 12 and np.array_equal(d2.basis_fom, basis[:, :12])
    assert list(d2.dofs) == list(d.dofs)[:12]                    # greedy is nested in the leading columns
    rows = [t[0] for t in d2.dofs]                               # store_dofs keeps (dof,) tuples (deim.py:217-224)
    assert_allclose(d2.PT_U, basis[rows, :12], rtol=0, atol=0)
    dump_pickle(StorageNames.ROM, basis)
    assert np.array_equal(read_pickle(StorageNames.ROM), basis)
    with open(StorageNames.MU_SPACE, "w") as fp:
        json.dump({"offline": [dict(alpha=1.0)], "online": [], "validation": []}, fp)
    assert json.load(open(StorageNames.MU_SPACE))["offline"][0]["alpha"] == 1.0


# ----------------------------------------------------------------------------------------------
# host-logic runs (device operators stubbed) ---------------------------------------------------
def test_orth_golden_hostlogic(cpu_ops, golden_orth):
    check_orth_golden(golden_orth)


def test_orth_semantics_hostlogic(cpu_ops):
    check_orth_semantics()


def test_deim_golden_hostlogic(cpu_ops, golden_deim):
    check_deim_golden(golden_deim)


@pytest.mark.parametrize("operator", ["stiffness", "mass", "convection", "stiffness_ale"])
def test_mdeim_end_to_end_hostlogic(cpu_ops, operator):
    check_mdeim_end_to_end(operator)


def test_deim_vector_hostlogic(cpu_ops):
    check_deim_vector_end_to_end()


@pytest.mark.parametrize("case", ["r10_bdf1", "r10_bdf2", "r24_bdf1", "r24_bdf2"])
def test_rom_online_hostlogic(cpu_ops, golden_rom, case):
    check_rom_online_golden(golden_rom, case)


def test_rom_offline_hostlogic(cpu_ops):
    check_rom_offline_and_hyper_reduced()


def test_to_rom_hostlogic(cpu_ops):
    check_to_rom_roundtrip()


def test_hyper_reduced_loop_hostlogic(cpu_ops):
    check_hyper_reduced_loop_matches_oracle(device_sweep=False)


def test_artefact_round_trip_hostlogic(cpu_ops, golden_deim, tmp_path, monkeypatch):
    check_artefact_round_trip(golden_deim, tmp_path, monkeypatch)


def test_linear_rom_classes_heat_hostlogic(cpu_ops, golden_heat):
    check_linear_rom_classes_on_the_heat_problem(golden_heat)


def test_product_path_fails_loudly_without_gpu():
    """No silent CPU fallback: on a box without a GPU the hot path raises."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from romtime_amd import RomtimeHipError, orth

    with pytest.raises((RomtimeHipError, RuntimeError, AssertionError)):
        orth(np.ones((8, 2)))


# ----------------------------------------------------------------------------------------------
# HIP runs ---------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_orth_golden_hip(golden_orth):
    check_orth_golden(golden_orth)


@pytest.mark.gpu
def test_orth_semantics_hip():
    check_orth_semantics()


@pytest.mark.gpu
def test_deim_golden_hip(golden_deim):
    check_deim_golden(golden_deim)


@pytest.mark.gpu
@pytest.mark.parametrize("operator", ["stiffness", "mass", "convection", "stiffness_ale"])
def test_mdeim_end_to_end_hip(operator):
    check_mdeim_end_to_end(operator)


@pytest.mark.gpu
def test_deim_vector_hip():
    check_deim_vector_end_to_end()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["r10_bdf1", "r10_bdf2", "r24_bdf1", "r24_bdf2"])
def test_rom_online_hip(golden_rom, case):
    check_rom_online_golden(golden_rom, case)


@pytest.mark.gpu
def test_rom_offline_hip():
    check_rom_offline_and_hyper_reduced()


@pytest.mark.gpu
def test_to_rom_hip():
    check_to_rom_roundtrip()


@pytest.mark.gpu
def test_hyper_reduced_loop_hip():
    check_hyper_reduced_loop_matches_oracle(device_sweep=True)


@pytest.mark.gpu
def test_linear_rom_classes_heat_hip(golden_heat):
    check_linear_rom_classes_on_the_heat_problem(golden_heat)


@pytest.mark.gpu
def test_artefact_round_trip_hip(golden_deim, tmp_path, monkeypatch):
    check_artefact_round_trip(golden_deim, tmp_path, monkeypatch)   # load_fom_basis(keep=) re-runs the device greedy (a14)


def check_orth_odd_shapes():
    """Shapes and inputs at the edges of the device path (n < 3 uses the host eigensolver, F-ordered views,
    CUDA tensors in -> CUDA tensor out, all-zero snapshots), each against the oracle."""
    import torch

    from romtime_amd import orth

    rng = np.random.RandomState(11)
    for N, n, kw in [(5, 2, {}), (50, 1, {}), (40, 60, dict(num=60)), (12, 30, {}), (1000, 3, dict(num=2)), (37, 5, dict(tol=0.99)), (300, 17, dict(num=40)),
                     (2000, 64, dict(num=10)), (9000, 130, dict(num=12, normalize=False))]:
        X = rng.standard_normal((N, n)) * 10.0 ** (-0.3 * np.arange(n))
        for arr in (np.ascontiguousarray(X), np.asfortranarray(X)):
            Q, s, e = orth(arr, **kw)
            Qr, sr, er = oracle.orth(X.copy(), **kw)
            assert Q.shape == Qr.shape and s.shape == sr.shape
            assert np.all(np.abs(s - sr) <= 2e-13 * sr[0] + 8 * EPS * sr[0] ** 2 / np.maximum(sr, 1e-300))
            if Q.shape[1]:
                assert np.linalg.norm(Q @ (Q.T @ Qr) - Qr, 2) < 1e-9
    # all-zero snapshots: energy is 0/0 = NaN in the reference as well; nothing is kept
    Z = np.zeros((20, 4))
    Q, s, e = orth(Z, normalize=False)
    Qr, sr, er = oracle.orth(Z, normalize=False)
    assert Q.shape == Qr.shape == (20, 0) and np.all(s == 0) and np.all(np.isnan(e)) and np.all(np.isnan(er))
    # ``num`` above the numerical rank with an exactly zero tail (a forcing that vanishes at several times, walked
    # with normalize=False and num_t, deim.py:390-395): every returned column is finite; the kept directions match
    # the oracle's and the columns beyond the rank are zero columns on both routes (dgesvd completes the basis with
    # arbitrary orthonormal vectors there; no caller can depend on them)
    R = np.zeros((60, 5))
    R[:, 0] = rng.standard_normal(60)
    R[:, 3] = 1e-3 * rng.standard_normal(60)
    for passes in (1, "deflate"):
        Q, s, e = orth(R, num=4, normalize=False, passes=passes)
        Qr, sr, er = oracle.orth(R, num=4, normalize=False)
        assert Q.shape == (60, 4) and np.all(np.isfinite(Q)), passes
        assert np.all(np.abs(s - sr) <= 2e-13 * sr[0] + 8 * EPS * sr[0] ** 2 / np.maximum(sr, 1e-300))  # the POD bar
        assert np.all(s[2:] <= 1e-7 * sr[0])     # exact zeros come back as Gram rounding, sqrt(eps) sigma_1 at most
        for i in range(2):
            assert min(np.linalg.norm(Q[:, i] - Qr[:, i]), np.linalg.norm(Q[:, i] + Qr[:, i])) < 1e-9, (passes, i)
        assert np.abs(Q[:, 2:]).max() < 1e-12, passes     # zero columns (rounding residue divided by a noise sigma at most)
    from romtime_amd import ops as _ops

    if torch.cuda.is_available() and _ops.to_device.__module__ == _ops.__name__:   # device operators, not the host stand-ins
        Xd = torch.from_numpy(rng.standard_normal((500, 8))).cuda()
        Qd, s, e = orth(Xd, num=3)
        assert isinstance(Qd, torch.Tensor) and Qd.is_cuda and Qd.shape == (500, 3)


def test_orth_odd_shapes_hostlogic(cpu_ops):
    check_orth_odd_shapes()


@pytest.mark.gpu
def test_orth_odd_shapes_hip():
    check_orth_odd_shapes()


@pytest.mark.gpu
def test_rt_pod_orth_composite_abi(golden_orth):
    """The C entry point rt_pod_orth (the whole of orth for hosts without the Python layer) on the 48 reference-generated
    cases - the bars of check_orth_golden - plus deep spectra (deflated levels inside the call), clustered eigenvalues
    (Rayleigh-Ritz on the host's k x k problem), n < 3 (host Jacobi), more snapshots than DoFs, a too-small Q, zero norms."""
    import torch

    from romtime_amd import ops
    from romtime_amd._lib import RomtimeHipError

    g = golden_orth
    for key in g["cases"]:
        mname, bname, nname = str(key).split("__")
        kw = _branch_kwargs(bname)
        X = g[f"X__{mname}"]
        Qd, s, energy, levels = ops.pod_orth(ops.to_device(X), normalize=(nname == "norm"), **kw)
        Q = Qd.cpu().numpy()
        gQ, gs, ge = g[f"Q__{key}"], g[f"s__{key}"], g[f"energy__{key}"]
        knife_edge = "tol" in kw and np.abs(ge - kw["tol"]).min() < 1e-12
        if knife_edge:
            assert abs(Q.shape[1] - gQ.shape[1]) <= 1
            keep = min(Q.shape[1], gQ.shape[1])
            Q, gQ = Q[:, :keep], gQ[:, :keep]
        assert Q.shape == gQ.shape and s.shape == gs.shape, key
        assert np.all(np.abs(s - gs) <= 2e-13 * gs[0] + 8 * EPS * gs[0] ** 2 / np.maximum(gs, 1e-300)), key
        assert_allclose(energy, ge, rtol=1e-10, atol=0, err_msg=str(key))
        for i in range(Q.shape[1]):
            err = min(np.linalg.norm(Q[:, i] - gQ[:, i]), np.linalg.norm(Q[:, i] + gQ[:, i]))
            assert err <= pod_column_tolerance(gs, i), (key, i, err, levels)
    rng = np.random.RandomState(5)
    # a 4-fold cluster of leading singular values: individual vectors are arbitrary, the span is not
    U, _ = np.linalg.qr(rng.standard_normal((3000, 20)))
    V, _ = np.linalg.qr(rng.standard_normal((20, 20)))
    X = (U * np.r_[np.full(4, 3.0), 10.0 ** (-0.2 * np.arange(16))]) @ V.T
    Qd, s, e, _ = ops.pod_orth(ops.to_device(X), num=4, normalize=False)
    Qo, so, _ = oracle.orth(X, num=4, normalize=False)
    Q = Qd.cpu().numpy()
    assert np.linalg.norm(Q @ (Q.T @ Qo) - Qo, 2) < 1e-10 and np.abs(Q.T @ Q - np.eye(4)).max() < 1e-12
    for N, n, kw in [(5, 2, {}), (50, 1, {}), (40, 60, dict(num=60)), (12, 30, {}), (9000, 130, dict(num=12, normalize=False))]:
        X = rng.standard_normal((N, n)) * 10.0 ** (-0.3 * np.arange(n))
        for arr in (np.ascontiguousarray(X), np.asfortranarray(X)):
            Qd, s, e, _ = ops.pod_orth(ops.to_device(arr), **kw)
            Qr, sr, er = oracle.orth(X.copy(), **kw)
            assert Qd.shape == Qr.shape and s.shape == sr.shape, (N, n)
            assert np.all(np.abs(s - sr) <= 2e-13 * sr[0] + 8 * EPS * sr[0] ** 2 / np.maximum(sr, 1e-300))
            Q = Qd.cpu().numpy()
            if Q.shape[1]:
                assert np.linalg.norm(Q @ (Q.T @ Qr) - Qr, 2) < 1e-9
    with pytest.raises(RomtimeHipError):                       # the rule keeps 6 modes, Q holds 2: RT_ERR_ARG
        ops.pod_orth(ops.to_device(rng.standard_normal((200, 6))), q_cols=2)
    Z = rng.standard_normal((50, 6))
    Z[:, 2] = 0.0
    with pytest.raises(ValueError):
        ops.pod_orth(ops.to_device(Z), normalize=True)
    Xd = torch.from_numpy(rng.standard_normal((500, 8))).cuda()
    Qd, s, e, levels = ops.pod_orth(Xd, num=3)
    assert Qd.is_cuda and Qd.shape == (500, 3) and levels == 1
