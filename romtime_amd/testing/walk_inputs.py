"""Seeded inputs of the tree-walk fixtures (tests/golden/walks.npz): the closed-form P1 FOM, the parameter points and
the operator callbacks that BOTH the fixture generator (tests/golden/make_golden.py, which hands them to the
reference's classes) and the tests (which hand them to romtime_amd's classes) use.  Inputs only - no reference code."""
import numpy as np

WALK_MUS = [dict(alpha_0=0.4, beta=2.0, delta=0.3), dict(alpha_0=1.1, beta=5.5, delta=1.2),
            dict(alpha_0=1.7, beta=8.0, delta=0.7), dict(alpha_0=0.9, beta=3.3, delta=1.8)]
WALK_TS = np.linspace(0.2, 5.0, 9)


def walk_forcing(x, t, **mu):
    """A forcing with genuinely (mu, t)-dependent shape: the per-mu time bases span different subspaces, so the
    mu-level spectrum has no exact multiplicities (the separable forcing of tests/test_deim.py gives sigma = sqrt(n_mu)
    twice, and then any rotation of the basis is an equally valid answer)."""
    return (np.sin(mu["beta"] * x * (1.0 + 0.2 * mu["delta"] * t)) * np.exp(-0.3 * t)
            + mu["alpha_0"] * np.cos(3.0 * x + mu["delta"] * t) + 0.5 * mu["delta"] * x * x * t)


def walk_solver():
    from .mock import MockBurgers

    fom = MockBurgers(domain=dict(L0=1.0, nx=60, T=5.0, nt=20), Lt=lambda t, **mu: 1.0 + 0.1 * mu["delta"] * t)
    fom.forcing_term = walk_forcing
    fom.setup()
    return fom


def walk_rich_operator(fom):
    """(mu, t)-dependent combination of the P1 operators (all on the tridiagonal pattern)."""

    def assemble(mu=None, t=None, entries=None):
        w = np.sin((1.0 + mu["delta"]) * np.pi * fom.x_at(mu, t) * (1.0 + 0.3 * t)) * mu["alpha_0"]
        parts = [(1.0, fom.assemble_stiffness(mu, t, entries=entries)),
                 (1.0 + t * mu["beta"], fom.assemble_mass(mu, t, entries=entries)),
                 (mu["delta"], fom.assemble_convection(mu, t, entries=entries)),
                 (1.0, fom.assemble_trilinear(mu, t, w, entries=entries))]
        out = parts[0][1] * parts[0][0]
        for c, a in parts[1:]:
            out = out + c * a
        if entries is not None:                      # Dirichlet rows: the combination would give 1+..., keep 1 / 0
            for k, (i, j) in enumerate(entries):
                if i in (0, fom.Nh - 1):
                    out[k] = 1.0 if i == j else 0.0
            return out
        out = out.tolil()
        for i in (0, fom.Nh - 1):
            out[i, :] = 0.0
            out[i, i] = 1.0
        return out.tocsr()

    return assemble


def walk_state_operator(fom):
    """State-dependent operator with a (mu, t)-dependent shape: the trilinear form of the weighted state
    u_n (1 + delta sin(beta t / 5) x) plus the nonlinear lifting.  (The bare 1-D P1 trilinear form does not depend on
    the mesh size, hence not on (mu, t): every level of the N-MDEIM walk then stacks copies of ONE subspace and all
    kept singular values coincide.)"""

    def assemble(mu=None, t=None, u_n=None, entries=None):
        mu2 = dict(mu, omega=1.0 + mu["beta"])
        x = np.linspace(0.0, 1.0, fom.Nh)
        w = np.asarray(u_n) * (1.0 + mu["delta"] * np.sin(mu["beta"] * t / 5.0) * x)
        a = fom.assemble_trilinear(mu2, t, w, entries=entries)
        b = fom.assemble_nonlinear_lifting(mu2, t, entries=entries)
        if entries is not None:
            out = a + b
            for k, (i, j) in enumerate(entries):
                if i in (0, fom.Nh - 1):
                    out[k] = 1.0 if i == j else 0.0
            return out
        out = (a + b).tolil()
        for i in (0, fom.Nh - 1):
            out[i, :] = 0.0
            out[i, i] = 1.0
        return out.tocsr()

    return assemble


RB_MUS = [dict(alpha_0=0.05 + 0.03 * i, delta=0.3 + 0.1 * i, omega=9.0 + 1.5 * i) for i in range(3)]
RB_CASES = (("rb_default", {}, None), ("rb_tol", {"tol_time": 1.0 - 1e-11, "tol_mu": 1.0 - 1e-10}, None), ("rb_num", {}, 9))


def rb_fom():
    """The Burgers-type FOM whose solves feed build_reduced_basis in the fixtures."""
    from .mock import MockBurgers

    fom = MockBurgers(domain=dict(L0=1.0, nx=60, T=0.55, nt=22), Lt=lambda t, **mu: 1.0 - 0.1 * np.sin(mu["omega"] * t),
                      bdf2=True)
    fom.setup()
    return fom


def nmdeim_states(Nh):
    """Four smooth state functions (the role the reduced basis plays in N-MDEIM.run, nonlinear.py:159-170)."""
    rng = np.random.RandomState(5)
    x = np.linspace(0.0, 1.0, Nh)
    psi = np.array([np.sin((k + 1) * np.pi * x) * (1.0 + 0.3 * k * x) for k in range(4)]).T
    psi += 1e-2 * rng.standard_normal(psi.shape)
    psi[0, :] = psi[-1, :] = 0.0
    return psi


# ---- piston workflow (rom/hrom.py:979-1182 call sequence) -------------------------------------------------------
PISTON_MUS = [dict(a0=18.0, omega=9.0, delta=0.30, alpha_0=0.05), dict(a0=15.0, omega=10.5, delta=0.40, alpha_0=0.08),
              dict(a0=14.0, omega=12.0, delta=0.45, alpha_0=0.11)]
PISTON_TS = np.linspace(0.025, 0.55, 8)
PISTON_SROM_TRUNCATE = 2
PISTON_TOL_TIME, PISTON_TOL_MU = 1.0 - 1e-4, 1.0 - 1e-6   # a coarse S-ROM (6 modes) / ROM (4 modes) pair: errors of 1e-3, not rounding


def piston_grid():
    """Distributions of the piston parameters (a0, omega, delta drive the Mach-number sampler of
    RomConstructorNonlinear.build_sampling_space, rom.py:760-860; alpha_0 is the mock's viscosity)."""
    from scipy.stats.distributions import uniform

    return {"a0": uniform(10.0, 10.0), "omega": uniform(8.0, 4.0), "delta": uniform(0.2, 0.3), "alpha_0": uniform(0.04, 0.08)}


def heat_problem(moving, nx=160, nt=40, r=9):
    """Config 1's problem - the manufactured heat equation MFP1 (problems/mfp1.py:19-75) on a fixed or moving interval -
    with a fixed orthonormal trial basis (low sine modes, homogeneous at the Dirichlet ends) and the (mu, t) states at
    which tests/golden/make_golden.py::gen_heat records the reference's reduced operators.  Returns (fom, V, states)."""
    from .mock import MockHeatEquation

    kw = {}
    if moving:
        kw = dict(Lt=lambda t, **mu: 1.0 - 0.25 * np.sin(mu["omega"] * t),
                  dLt_dt=lambda t, **mu: -0.25 * mu["omega"] * np.cos(mu["omega"] * t))
    fom = MockHeatEquation(domain=dict(L0=2.0, nx=nx, T=1.0, nt=nt), **kw)
    fom.setup()
    rng = np.random.RandomState(7)
    x = np.linspace(0.0, 1.0, fom.Nh)
    modes = np.array([np.sin((k + 1) * np.pi * x) for k in range(r)]).T + 1e-3 * rng.standard_normal((fom.Nh, r))
    modes[0, :] = modes[-1, :] = 0.0
    V, _ = np.linalg.qr(modes)
    states = [(dict(delta=0.4 + 0.3 * q, beta=2.0 + 1.5 * q, alpha_0=0.3 + 0.25 * q, omega=1.0 + q), 0.05 + 0.21 * q)
              for q in range(4)]
    return fom, V, states
