"""Tree walks, truncation and reduced-basis construction against fixtures produced by the
REFERENCE'S OWN ``DiscreteEmpiricalInterpolation.run`` / ``MatrixDiscreteEmpiricalInterpolation.run`` /
``MatrixDiscreteEmpiricalInterpolationNonlinear.run`` + ``truncate`` / ``RomConstructorNonlinear.build_reduced_basis`` +
``truncate`` (tests/golden/make_golden.py::gen_walks -> walks.npz), and the product's small utilities (a12 / a13) against
the ``ez_* / err / diff`` fixtures of deim.npz.

What can be compared.  A tree walk stacks per-parameter orthonormal bases and takes their POD (deim.py:340-349,
rom.py:366-384): directions shared by several parameters give EXACTLY equal singular values (sqrt(n_mu), visible in
every fixture), and inside such a cluster any rotation is an equally valid answer - the reference's own columns there
are whatever LAPACK happened to return.  Moreover a level's output depends on the previous level only through the
SUBSPACES kept there, which are determined to eps sigma_1 / (gap at the truncation boundary).  So the checks are:

  * per-level kept-mode counts, the bookkeeping of ``report``: identical;
  * spectra: within the POD bar of tests/test_surface.py plus the Weyl bound for the level's input perturbation;
  * bases: the span of the leading k columns at every k where the reference spectrum has a gap (Davis-Kahan bound
    ``tol_k = 1e-9 + (E_in + 20 eps sigma_1) / (sigma_k - sigma_k+1)`` with ``E_in`` the input perturbation of the level);
  * interpolation indices: equal to the reference's when the spectrum has no cluster and the reference's own greedy
    margins exceed the basis tolerance; otherwise equal to the oracle's greedy on the product's own basis.
Both runs: host logic on the CPU stub (here) and the HIP path (-m gpu)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import romtime_oracle as oracle

EPS = 2.2e-16
MUS_KEYS = ("alpha_0", "beta", "delta")


C_BW = 20.0   # backward-error constant: a stable SVD returns the exact factors of A + E with ||E|| <= C_BW eps sigma_1


def _boundary_error(s, kept):
    """C eps sigma_1 / (sigma_kept - sigma_kept+1), sigma_n+1 := 0: how well a truncated POD determines the span it
    keeps (Wedin).  With every mode kept the span is the column space, determined to eps sigma_1 / sigma_min."""
    s = np.asarray(s)
    if kept == 0:
        return C_BW * EPS
    nxt = s[kept] if kept < len(s) else 0.0
    return C_BW * EPS * s[0] / max(s[kept - 1] - nxt, 1e-300)


def _span_check(Q, Qref, s_ref, E_in, label):
    assert Q.shape == Qref.shape, (label, Q.shape, Qref.shape)
    n = Q.shape[1]
    worst = 0.0
    for k in range(1, n + 1):
        nxt = s_ref[k] if k < len(s_ref) else 0.0
        gap = s_ref[k - 1] - nxt
        if gap < 1e-6 * s_ref[0] and k < n:
            continue                                           # inside a cluster of (nearly) equal singular values
        A, B = Q[:, :k], Qref[:, :k]
        dist = np.linalg.norm(A - B @ (B.T @ A), 2)
        tol = 1e-9 + (E_in + C_BW * EPS * s_ref[0]) / gap
        assert dist <= tol, (label, k, dist, tol)
        worst = max(worst, dist / tol)
    return worst


def _spectrum_check(s, s_ref, E_in, label):
    s, s_ref = np.asarray(s), np.asarray(s_ref)
    assert s.shape == s_ref.shape, (label, s.shape, s_ref.shape)
    bar = 2e-13 * s_ref[0] + 8 * EPS * s_ref[0] ** 2 / np.maximum(s_ref, 1e-300) + E_in * s_ref[0]
    assert np.all(np.abs(s - s_ref) <= bar), (label, np.abs(s - s_ref).max())


def _has_cluster(s_ref, kept):
    k = np.asarray(s_ref)[: kept + 1]
    return bool(np.any(np.abs(np.diff(k)) < 1e-6 * k[0]))


def _mus(g, key="mus", names=MUS_KEYS):
    return [dict(zip(names, row)) for row in g[key]]


def _level_input_error(g, key, n_mu):
    return max(_boundary_error(g[f"{key}__spectrum_time_{i}"], int(g[f"{key}__basis_time"][i])) for i in range(n_mu))


def _check_report(red, g, key, n_mu):
    from romtime_amd.conventions import Stage

    off = red.report[Stage.OFFLINE]
    assert [off["basis-shape-time"][i] for i in range(n_mu)] == list(g[f"{key}__basis_time"]), key
    assert off["basis-shape-after-tree-walk"] == int(g[f"{key}__after_walk"]), key
    assert off["basis-shape-final"] == int(g[f"{key}__final"]) == red.N, key
    E_in = _level_input_error(g, key, n_mu)
    for i in range(n_mu):
        _spectrum_check(off["spectrum-time"][i], g[f"{key}__spectrum_time_{i}"], 0.0, (key, "time", i))
    _spectrum_check(off["spectrum-mu"], g[f"{key}__spectrum_mu"], E_in, (key, "mu"))
    assert_allclose(off["energy-mu"], g[f"{key}__energy_mu"], rtol=1e-8, atol=0, err_msg=key)
    return E_in


def _check_dofs(red, g, key, E_in, flat):
    """``flat(dofs)`` maps the reductor's dof tuples to indices into its value vector."""
    s_ref = g[f"{key}__spectrum_mu"]
    mine = flat(red.dofs)
    own, PT_U_own, _ = oracle.deim_greedy(red.basis_fom)
    assert list(mine) == list(own), key                      # the product's greedy == deim.py:517-561 on its own basis
    np.testing.assert_array_equal(red.PT_U, PT_U_own)
    margin = g[f"{key}__margin"] if f"{key}__margin" in g else None
    gaps = np.abs(np.diff(np.asarray(s_ref)[: red.N + 1]))
    basis_tol = 1e-9 + (E_in + C_BW * EPS * s_ref[0]) / max(gaps.min(), 1e-300) if gaps.size else 1e-9
    exact = (not _has_cluster(s_ref, red.N)) and margin is not None and basis_tol < 1e-3 * margin.min()
    if exact:
        assert list(mine) == list(flat_ref(g, key)), key
    return exact


def flat_ref(g, key):
    d = g[f"{key}__dofs"]
    if d.ndim == 1:
        return list(d)
    lut = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(g[f"{key}__rows"], g[f"{key}__cols"]))}
    return [lut[(int(r), int(c))] for r, c in d]


def _walk_fom():
    from romtime_amd.testing.walk_inputs import walk_solver

    return walk_solver()


def check_deim_walks(g):
    """DiscreteEmpiricalInterpolation.run (deim.py:175-215, 279-397): default / num / tol truncation."""
    from romtime_amd import DiscreteEmpiricalInterpolation

    fom = _walk_fom()
    mus, ts = _mus(g), g["ts"]
    compared_exactly = []
    for key, extra in (("deim_default", {}), ("deim_num", {"num_mu": 6, "num_time": 4}),
                       ("deim_tol", {"tol_mu": 1.0 - 1e-9, "tol_time": 1.0 - 1e-10})):
        d = DiscreteEmpiricalInterpolation(assemble=fom.assemble_forcing, grid=None,
                                           tree_walk_params=dict({"ts": ts, "num_snapshots": len(mus)}, **extra), name=key)
        d.setup(rnd=np.random.RandomState(0))
        d.run(mu_space=[dict(m) for m in mus])
        E_in = _check_report(d, g, key, len(mus))
        assert_allclose(d.sigmas, d.report["offline"]["spectrum-mu"])
        _span_check(d.basis_fom, g[f"{key}__basis_fom"], g[f"{key}__spectrum_mu"], E_in, key)
        if _check_dofs(d, g, key, E_in, lambda dofs: [i for (i,) in dofs]):
            compared_exactly.append(key)
    assert "deim_num" in compared_exactly, compared_exactly     # at least the well-conditioned walk pins the indices


def check_mdeim_walks(g):
    """MatrixDiscreteEmpiricalInterpolation.run on the reference's acceptance operator (separable: a degenerate
    spectrum) and on a (mu, t)-rich combination (mdeim.py:102-151 topology, deim.py:388-389 zeroed row)."""
    from romtime_amd import MatrixDiscreteEmpiricalInterpolation
    from romtime_amd.base import Reductor
    from romtime_amd.testing.walk_inputs import walk_rich_operator

    fom = _walk_fom()
    mus, ts = _mus(g), g["ts"]
    for key, fn in (("mdeim_stiffness", fom.assemble_stiffness), ("mdeim_rich", walk_rich_operator(fom))):
        md = MatrixDiscreteEmpiricalInterpolation(assemble=fn, grid=None,
                                                  tree_walk_params={"ts": ts, "num_snapshots": len(mus)}, name=key)
        Reductor.setup(md, rnd=np.random.RandomState(0))
        md.rows, md.cols = md.get_matrix_topology(mu=mus[0], t=1.0)
        assert md.rows == list(g[f"{key}__rows"]) and md.cols == list(g[f"{key}__cols"])      # a4: topology
        md.run(mu_space=[dict(m) for m in mus])
        E_in = _check_report(md, g, key, len(mus))
        _span_check(md.basis_fom, g[f"{key}__basis_fom"], g[f"{key}__spectrum_mu"], E_in, key)
        lut = {(r, c): i for i, (r, c) in enumerate(zip(md.rows, md.cols))}
        _check_dofs(md, g, key, E_in, lambda dofs: [lut[tuple(rc)] for rc in dofs])
        assert np.all(md.basis_fom[0, :] == 0.0)                # the boundary entry never enters the basis


def check_nmdeim_walk_and_truncate(g):
    """N-MDEIM: three-level walk, all levels normalised (nonlinear.py:159-212, 320-468), then truncate(n)
    (nonlinear.py:49-104): leading columns kept, greedy re-run, report / name / topology carried over."""
    from romtime_amd import MatrixDiscreteEmpiricalInterpolationNonlinear
    from romtime_amd.conventions import Stage
    from romtime_amd.testing.walk_inputs import nmdeim_states, walk_state_operator

    fom = _walk_fom()
    mus, ts = _mus(g)[:3], g["ts"][::2]
    key = "nmdeim"
    x = np.linspace(0.0, 1.0, fom.Nh)
    nm = MatrixDiscreteEmpiricalInterpolationNonlinear(assemble=walk_state_operator(fom), grid=None,
                                                       tree_walk_params={"ts": ts, "num_snapshots": 3}, name=key)
    from romtime_amd.base import Reductor

    Reductor.setup(nm, rnd=np.random.RandomState(0))
    nm.rows, nm.cols = nm.get_matrix_topology(mu=mus[0], t=1.0, u_n=x)
    assert nm.rows == list(g[f"{key}__rows"]) and nm.cols == list(g[f"{key}__cols"])
    np.testing.assert_array_equal(nmdeim_states(fom.Nh), g[f"{key}__psi"])
    nm.run(u_n=g[f"{key}__psi"], mu_space=[dict(m) for m in mus])
    E_in = _check_report(nm, g, key, 3)
    # the time level of this walk is itself a two-level POD (over psi, then over t): its output carries the inner
    # level's boundary error as well; the fixture's spectra bound it by the same expression
    _span_check(nm.basis_fom, g[f"{key}__basis_fom"], g[f"{key}__spectrum_mu"], 10 * E_in, key)
    lut = {(r, c): i for i, (r, c) in enumerate(zip(nm.rows, nm.cols))}
    flat = lambda dofs: [lut[tuple(rc)] for rc in dofs]
    _check_dofs(nm, g, key, E_in, flat)
    n_cut = int(g[f"{key}__trunc_n"])
    tr = nm.truncate(n_cut)
    assert tr.name == str(g[f"{key}__trunc_name"]) and tr.N == nm.N - n_cut == int(g[f"{key}__trunc_final"])
    assert tr.report[Stage.OFFLINE]["basis-shape-final"] == tr.N and tr.rows is nm.rows
    np.testing.assert_array_equal(tr.basis_fom, nm.basis_fom[:, : tr.N])
    own, PT_U_own, _ = oracle.deim_greedy(tr.basis_fom)
    assert flat(tr.dofs) == list(own)
    np.testing.assert_array_equal(tr.PT_U, PT_U_own)
    assert tr.mu_space == nm.mu_space and tr.mu_space is not nm.mu_space
    with pytest.raises(AssertionError):
        nm.truncate(nm.N)
    # the interpolant at a training (mu, t) of a state inside the training span (FOM form, entry 0 pinned,
    # nonlinear.py:247-283): the reference's reductor and the product's were trained on the same span, so both
    # reproduce the assembled operator, and hence each other
    mu, u, t = _mus(g)[1], g[f"{key}__probe_u"], float(g[f"{key}__probe_t"])
    exact = oracle.eliminate_zeros(walk_state_operator(fom)(mu=mu, t=t, u_n=u)).data
    mine, ref_val = nm._interpolate(mu=mu, t=t, u_n=u, which=nm.FOM), g[f"{key}__interp_fom"]
    assert mine[0] == 1.0 and ref_val[0] == 1.0
    scale = np.abs(exact).max()
    assert np.abs(ref_val[1:] - exact[1:]).max() <= 1e-6 * scale
    assert np.abs(mine[1:] - exact[1:]).max() <= 1e-6 * scale


def check_reduced_basis_walks(g):
    """RomConstructorNonlinear.build_reduced_basis (rom.py:276-412: time level normalised, mu level not; the
    nonlinear-term walk beside it) and truncate (rom.py:169-198)."""
    from romtime_amd import RomConstructorNonlinear
    from romtime_amd.conventions import Stage
    from romtime_amd.testing.walk_inputs import RB_CASES, rb_fom

    for key, tolerances, num_basis in RB_CASES:
        fom = rb_fom()
        mus = _mus(g, f"{key}__mus", ("alpha_0", "delta", "omega"))
        rom = RomConstructorNonlinear(fom=fom, grid=None, name="S-ROM")
        rom.setup(rnd=0)
        sols = rom.build_reduced_basis(mu_space=[dict(m) for m in mus], num_basis=num_basis, tolerances=tolerances)
        assert_allclose(sols[1], g[f"{key}__fom_solution_1"], rtol=0, atol=1e-12)        # same FOM on both sides
        off = rom.report[Stage.OFFLINE]
        assert [off["basis-shape-time"][i] for i in range(3)] == list(g[f"{key}__basis_time"]), key
        assert off["basis-shape-after-tree-walk"] == int(g[f"{key}__after_walk"])
        assert off["basis-shape-final"] == int(g[f"{key}__final"]) == rom.N
        assert [off["N-basis-shape-time"][i] for i in range(3)] == list(g[f"{key}__N_basis_time"])
        assert off["N-basis-shape-after-tree-walk"] == int(g[f"{key}__N_after_walk"])
        assert off["N-basis-shape-final"] == int(g[f"{key}__N_final"]) == rom.basis_nonlinear.shape[1]
        E_in = max(_boundary_error(g[f"{key}__spectrum_time_{i}"], int(g[f"{key}__basis_time"][i])) for i in range(3))
        E_nl = max(_boundary_error(g[f"{key}__N_spectrum_time_{i}"], _kept_nonlinear(g, key, i, tolerances)) for i in range(3))
        for i in range(3):
            _spectrum_check(off["spectrum-time"][i], g[f"{key}__spectrum_time_{i}"], 0.0, (key, "time", i))
            _spectrum_check(off["N-spectrum-time"][i], g[f"{key}__N_spectrum_time_{i}"], 0.0, (key, "N time", i))
        _spectrum_check(off["spectrum-mu"], g[f"{key}__spectrum_mu"], E_in, (key, "mu"))
        _spectrum_check(off["N-spectrum-mu"], g[f"{key}__N_spectrum_mu"], E_nl, (key, "N mu"))
        _span_check(rom.basis, g[f"{key}__basis"], g[f"{key}__spectrum_mu"], E_in, key)
        _span_check(rom.basis_nonlinear, g[f"{key}__basis_nonlinear"], g[f"{key}__N_spectrum_mu"], E_nl, key + " nonlinear")
        tr = rom.truncate(2)
        assert tr.N == rom.N - 2 == int(g[f"{key}__trunc_final"]) and tr.report[Stage.OFFLINE]["basis-shape-final"] == tr.N
        np.testing.assert_array_equal(tr.basis, rom.basis[:, : tr.N])
        assert tr.mu_space == rom.mu_space and tr.fom is rom.fom
        with pytest.raises(AssertionError):
            rom.truncate(rom.N)


def _kept_nonlinear(g, key, i, tolerances):
    """Modes the nonlinear-term time level keeps (rom.py:343-352: same ``tol`` as the solution, else the drop rule)."""
    s = g[f"{key}__N_spectrum_time_{i}"]
    tol = tolerances.get("tol_time")
    if tol:
        e = np.cumsum(s ** 2) / np.sum(s ** 2)
        return int(np.count_nonzero(e < tol))
    return int(np.count_nonzero(s > 1e-7))


def check_small_utilities(g):
    """a12 / a13 on the PRODUCT functions: eliminate_zeros / vector_to_csr / bilinear_to_csr against the reference's
    outputs (utils.py:76-93,136-168), compute_error and compute_rom_difference (rom/base.py:52-73, utils.py:173-212),
    and the identities of the reference's tests/test_utils.py:6-40."""
    from scipy.sparse import csr_matrix

    from romtime_amd import utils
    from romtime_amd.base import Reductor

    A = csr_matrix((g["ez_data_in"].copy(), g["csr_indices"], g["csr_indptr"]))
    out = utils.eliminate_zeros(A)
    assert out is A                                                     # mutates and returns, like the reference
    np.testing.assert_array_equal(out.indptr, g["ez_indptr"])
    np.testing.assert_array_equal(out.indices, g["ez_indices"])
    np.testing.assert_array_equal(out.data, g["ez_data"])
    rows = np.repeat(np.arange(len(g["ez_indptr"]) - 1), np.diff(g["ez_indptr"]))
    back = utils.vector_to_csr(g["ez_data"], rows, g["ez_indices"])
    np.testing.assert_array_equal(back.toarray(), out.toarray())

    class PetscLike:                                                    # what a dolfin Matrix exposes (utils.py:90-91)
        def __init__(self, csr):
            self._c = csr

        def mat(self):
            return self

        def getValuesCSR(self):
            return self._c.indptr, self._c.indices, self._c.data

        size = property(lambda self: self._c.shape)

    B = csr_matrix((g["csr_data"], g["csr_indices"], g["csr_indptr"]))
    for form in (B, B.toarray(), PetscLike(B)):
        np.testing.assert_array_equal(utils.bilinear_to_csr(form).toarray(), B.toarray())
    assert utils.compute_error(g["err_u"], g["err_ue"]) == float(g["err"])
    assert Reductor._compute_error(g["err_u"], g["err_ue"]) == float(g["err"])
    assert_allclose(utils.compute_rom_difference(g["diff_uN"], g["diff_uNs"], g["diff_Vs"]), float(g["diff"]), rtol=1e-14)
    # tests/test_utils.py: equal ROM and S-ROM coefficients -> 0; an orthonormal S-ROM basis -> ||u_s - [u; 0]|| / sqrt(N)
    V, _ = np.linalg.qr(np.random.RandomState(0).standard_normal((30, 6)))
    u = np.arange(1.0, 5.0)
    assert utils.compute_rom_difference(u, np.r_[u, 0.0, 0.0], V) == 0.0
    us = np.r_[u, 0.5, -0.25]
    assert_allclose(utils.compute_rom_difference(u, us, V), np.sqrt(0.5 ** 2 + 0.25 ** 2) / np.sqrt(30), rtol=1e-14)
    assert_allclose(utils.project_csr(B, g["mdeim_V"]), g["project_csr"], rtol=0, atol=1e-13) if _have_ops() else None


def _have_ops():
    import torch

    from romtime_amd import ops

    return torch.cuda.is_available() or getattr(ops.project_csr, "__module__", "").endswith("cpu_stub")


@pytest.fixture(scope="module")
def golden_walks():
    from tests.conftest import load_golden

    return load_golden("walks.npz")


# ---- host logic (device operators stubbed with the oracle's arithmetic) ---------------------------------------
def test_deim_walks_hostlogic(cpu_ops, golden_walks):
    check_deim_walks(golden_walks)


def test_mdeim_walks_hostlogic(cpu_ops, golden_walks):
    check_mdeim_walks(golden_walks)


def test_nmdeim_walk_and_truncate_hostlogic(cpu_ops, golden_walks):
    check_nmdeim_walk_and_truncate(golden_walks)


def test_reduced_basis_walks_hostlogic(cpu_ops, golden_walks):
    check_reduced_basis_walks(golden_walks)


def test_small_utilities_hostlogic(cpu_ops, golden_deim):
    check_small_utilities(golden_deim)


# ---- HIP path ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_deim_walks_hip(golden_walks):
    check_deim_walks(golden_walks)


@pytest.mark.gpu
def test_mdeim_walks_hip(golden_walks):
    check_mdeim_walks(golden_walks)


@pytest.mark.gpu
def test_nmdeim_walk_and_truncate_hip(golden_walks):
    check_nmdeim_walk_and_truncate(golden_walks)


@pytest.mark.gpu
def test_reduced_basis_walks_hip(golden_walks):
    check_reduced_basis_walks(golden_walks)


@pytest.mark.gpu
def test_small_utilities_hip(golden_deim):
    check_small_utilities(golden_deim)
