#!/usr/bin/env python3
"""Generate ``tests/golden/*.npz`` by RUNNING THE REFERENCE'S OWN SOURCE.

Run in the build container only (``python tests/golden/make_golden.py``); it
reads ``/root/reference/src`` by path and is a no-op where that does not exist.
Nothing of the reference is copied: the fixtures hold seeded inputs, the
reference's outputs on them, and the library versions that produced them.

How the reference is loaded:
  * ``rom/pod.py`` needs only numpy+scipy and is loaded by file path;
  * the other modules import FEniCS at module scope, which is not installed, so
    inert placeholder modules named ``fenics``, ``dolfin.cpp.la`` (with empty
    ``Matrix``/``Vector`` classes), ``ujson``, ``seaborn``, ``adjustText`` are
    registered first -- no FEniCS function is ever called by the hot path;
  * two NumPy/SciPy API drifts are bridged here and only here:
    ``np.reshape(a=, newshape=)`` (deim.py:535) and ``gmres(tol=, maxiter=1e6)``
    (rom.py:36).
"""
import os
import sys
import types
import importlib.util

import numpy as np
import scipy
import scipy.sparse.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, REPO)


def _versions():
    return dict(numpy=np.__version__, scipy=scipy.__version__)


def load_reference():
    # --- placeholders for absent third-party modules -----------------------
    class Matrix:  # dolfin.cpp.la.Matrix stand-in: wraps a scipy CSR
        def __init__(self, csr):
            self._csr = csr.tocsr()

        # what bilinear_to_csr touches (utils.py:90-91)
        def mat(self):
            return self

        def getValuesCSR(self):
            c = self._csr
            return c.indptr, c.indices, c.data

        @property
        def size(self):
            return self._csr.shape

    class Vector:  # dolfin.cpp.la.Vector stand-in: wraps an ndarray
        def __init__(self, arr):
            self._arr = np.asarray(arr, dtype=float)

        def __array__(self, dtype=None, copy=None):
            return self._arr

        def __len__(self):
            return len(self._arr)

    fenics = types.ModuleType("fenics")
    fenics.as_backend_type = lambda m: m
    fenics.DOLFIN_EPS = 3e-16
    dolfin = types.ModuleType("dolfin")
    dolfin_cpp = types.ModuleType("dolfin.cpp")
    dolfin_la = types.ModuleType("dolfin.cpp.la")
    dolfin_la.Matrix = Matrix
    dolfin_la.Vector = Vector
    dolfin.cpp = dolfin_cpp
    dolfin_cpp.la = dolfin_la
    seaborn = types.ModuleType("seaborn")
    seaborn.set_theme = lambda *a, **k: None  # rom/hrom.py:42 calls it at import
    import json

    ujson = types.ModuleType("ujson")  # utils.py:214-225 dump_json / read_json: same wire format as json
    ujson.dump, ujson.load = json.dump, json.load
    for name, mod in [("fenics", fenics), ("dolfin", dolfin), ("dolfin.cpp", dolfin_cpp),
                      ("dolfin.cpp.la", dolfin_la), ("ujson", ujson), ("seaborn", seaborn)]:
        sys.modules[name] = mod
    adj = types.ModuleType("adjustText")
    adj.adjust_text = lambda *a, **k: None
    sys.modules["adjustText"] = adj

    # --- API drift shims (generator only) ----------------------------------
    _reshape = np.reshape

    def reshape(*args, **kwargs):
        if "a" in kwargs:
            args = (kwargs.pop("a"),) + args
        if "newshape" in kwargs:
            kwargs["shape"] = kwargs.pop("newshape")
        return _reshape(*args, **kwargs)

    np.reshape = reshape
    _gmres = scipy.sparse.linalg.gmres

    def gmres(A, b, tol=None, maxiter=None, **kw):
        if tol is not None:
            kw["rtol"] = tol
        if maxiter is not None:
            maxiter = int(maxiter)
        return _gmres(A, b, maxiter=maxiter, **kw)

    scipy.sparse.linalg.gmres = gmres

    sys.path.insert(0, REF)
    import matplotlib

    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("ref_pod", os.path.join(REF, "romtime/rom/pod.py"))
    ref_pod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_pod)
    import romtime.deim as ref_deim
    import romtime.rom.rom as ref_rom
    import romtime.rom.base as ref_base
    import romtime.utils as ref_utils
    import romtime.deim.nonlinear as ref_nonlinear
    import romtime.rom.hrom as ref_hrom
    import romtime.conventions as ref_conv

    return types.SimpleNamespace(pod=ref_pod, deim=ref_deim, rom=ref_rom, base=ref_base, nonlinear=ref_nonlinear,
                                 hrom=ref_hrom, conv=ref_conv, utils=ref_utils, Matrix=Matrix, Vector=Vector)


# ---------------------------------------------------------------------------
# seeded inputs (shared with the tests through the .npz files themselves)
# ---------------------------------------------------------------------------
def spectrum_matrix(rng, N, n, sigmas):
    U0, _ = np.linalg.qr(rng.standard_normal((N, n)))
    V0, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (U0 * sigmas) @ V0.T


def smooth_snapshots(N, n):
    """Config-1-like smooth parametrised snapshots (fast singular value decay)."""
    x = np.linspace(0.0, 1.0, N)
    ts = np.linspace(0.05, 5.0, n)
    cols = [(1.0 - np.exp(-0.7 * t)) * (1.0 + (0.3 + 0.1 * t) * x * x) * np.sin(np.pi * x * (1 + 0.05 * t))
            for t in ts]
    return np.array(cols).T


def gen_orth(ref, out):
    rng = np.random.RandomState(20260104)
    mats = {}
    mats["random_120x10"] = rng.standard_normal((120, 10))
    mats["decay_200x16"] = spectrum_matrix(rng, 200, 16, 10.0 ** (-np.arange(16) * 0.5))
    m = spectrum_matrix(rng, 160, 14, np.r_[10.0 ** (-np.arange(7) * 0.5), 10.0 ** (-9.0 - np.arange(7) * 0.5)])
    mats["gap_1e-7_160x14"] = m  # singular values straddle DROP_TOLERANCE with a clear gap
    z = spectrum_matrix(rng, 129, 12, 10.0 ** (-np.arange(12) * 0.4))
    z[0, :] = 0.0  # MDEIM convention: row 0 zeroed (deim.py:388-389)
    mats["zero_row_129x12"] = z
    mats["smooth_401x40"] = smooth_snapshots(401, 40)
    # an F-ordered view, as np.array(list_of_vectors).T produces (deim.py:384)
    mats["fview_90x8"] = np.array([rng.standard_normal(90) for _ in range(8)]).T
    branches = {
        "drop": dict(),
        "num": dict(num=5),
        "tol": dict(tol=1.0 - 1e-6),
        "tol_num": dict(tol=0.999, num=3),  # tol has precedence (pod.py:46-53)
    }
    data = {}
    cases = []
    for mname, X in mats.items():
        data[f"X__{mname}"] = X
        for bname, kw in branches.items():
            for normalize in (True, False):
                Q, s, energy, VT = ref.pod.orth(X.copy(), normalize=normalize, return_VT=True, **kw)
                key = f"{mname}__{bname}__{'norm' if normalize else 'raw'}"
                cases.append(key)
                data[f"Q__{key}"] = Q
                data[f"s__{key}"] = s
                data[f"energy__{key}"] = energy
                data[f"VT__{key}"] = VT
    data["cases"] = np.array(cases)
    data["versions"] = np.array(repr(_versions()))
    np.savez_compressed(os.path.join(out, "orth.npz"), **data)
    print("orth.npz:", len(cases), "cases")


def mdeim_p1_basis(ref):
    """Collateral basis of P1 stiffness+mass value-vectors on a tridiagonal pattern."""
    from romtime_amd.testing.mock import MockBurgers

    fom = MockBurgers(domain=dict(L0=1.0, nx=60, T=5.0, nt=20), Lt=lambda t, **mu: 1.0 + 0.2 * t * mu["delta"])
    fom.setup()
    rng = np.random.RandomState(3)
    snaps = []
    for _ in range(12):
        mu = dict(alpha_0=rng.uniform(0.01, 2.0), delta=rng.uniform(0.01, 2.0))
        for t in np.linspace(0.1, 5.0, 6):
            w = np.sin((1.0 + mu["delta"]) * np.pi * fom.x_at(mu, t) * (1.0 + 0.3 * t)) * mu["alpha_0"]
            A = (fom.assemble_stiffness(mu, t) + fom.assemble_mass(mu, t) * (1 + t)
                 + mu["delta"] * fom.assemble_convection(mu, t) + fom.assemble_trilinear(mu, t, w))
            snaps.append(A.data.copy())
    S = np.array(snaps).T
    S[0, :] = 0.0
    basis, _, _ = ref.pod.orth(S, normalize=False)
    A = fom.assemble_stiffness(dict(alpha_0=1.0, delta=1.0), 1.0)
    rows, cols = A.nonzero()
    order = np.argsort(rows, kind="stable")
    return basis, rows[order], cols[order], fom


def greedy_margins(basis, dofs):
    """(top1 - top2) / top1 of |residual| at every greedy step (0 = exact tie), from the reference's own selection."""
    m = np.empty(len(dofs))
    for k in range(len(dofs)):
        if k == 0:
            r = basis[:, 0]
        else:
            c = np.linalg.solve(basis[dofs[:k], :k], basis[dofs[:k], k])
            r = basis[:, k] - basis[:, :k] @ c
        a = np.sort(np.abs(r))[::-1]
        m[k] = (a[0] - a[1]) / a[0]
    return m


def gen_deim(ref, out):
    rng = np.random.RandomState(20260105)
    DEIM = ref.deim.DiscreteEmpiricalInterpolation
    MDEIM = ref.deim.MatrixDiscreteEmpiricalInterpolation
    data = {}

    def greedy(basis):
        d = DEIM(assemble=None, name="golden")
        d.basis_fom = basis
        dofs, P = d.build_interpolation_mesh()
        return np.array(dofs, dtype=np.int64), np.matmul(P.T, basis)

    def margins(basis, dofs):
        m = np.empty(len(dofs))
        for k in range(len(dofs)):
            if k == 0:
                r = basis[:, 0]
            else:
                c = np.linalg.solve(basis[dofs[:k], :k], basis[dofs[:k], k])
                r = basis[:, k] - basis[:, :k] @ c
            a = np.sort(np.abs(r))[::-1]
            m[k] = (a[0] - a[1]) / a[0]
        return m

    bases = {}
    bases["random_orth_300x16"], _ = np.linalg.qr(rng.standard_normal((300, 16)))
    # mirror-symmetric FE-like data: exact ties between i and N-1-i (SURVEY hard part C)
    x = np.linspace(0.0, 1.0, 201)
    sym = np.array([np.cos(2 * np.pi * k * (x - 0.5)) * np.exp(-k * 0.1) for k in range(1, 9)]).T
    bases["mirror_201x8"], _, _ = ref.pod.orth(sym, normalize=False)
    p1, rows, cols, fom = mdeim_p1_basis(ref)
    bases["mdeim_p1"] = p1
    bases["smooth_pod_401"] = ref.pod.orth(smooth_snapshots(401, 40), num=10)[0]
    for name, B in bases.items():
        B = np.ascontiguousarray(B)
        dofs, PT_U = greedy(B)
        data[f"basis__{name}"] = B
        data[f"dofs__{name}"] = dofs
        data[f"PT_U__{name}"] = PT_U
        data[f"margin__{name}"] = margins(B, dofs)
    data["names"] = np.array(list(bases))

    # --- project_basis (DEIM: deim.py:495-515; MDEIM: mdeim.py:153-192) --------
    V, _ = np.linalg.qr(rng.standard_normal((p1.shape[0] * 0 + fom.Nh, 8)))
    md = MDEIM(assemble=None, name="golden")
    md.basis_fom = p1[:, :5].copy()
    md.rows, md.cols = list(rows), list(cols)
    md.project_basis(V)
    data["mdeim_V"] = V
    data["mdeim_rows"] = rows.astype(np.int64)
    data["mdeim_cols"] = cols.astype(np.int64)
    data["mdeim_basis_fom"] = md.basis_fom
    data["mdeim_basis_rom"] = md.basis_rom
    data["mdeim_N_V"] = np.array(md.N_V)

    dd = DEIM(assemble=None, name="golden")
    dd.basis_fom = bases["random_orth_300x16"]
    V2, _ = np.linalg.qr(rng.standard_normal((300, 7)))
    dd.project_basis(V2)
    data["deim_V"] = V2
    data["deim_basis_rom"] = dd.basis_rom

    # --- _interpolate FOM / ROM form (deim.py:416-452, mdeim.py:230-261) -------
    md.load_fom_basis(basis=p1[:, :5].copy())  # re-runs greedy, stores dofs/PT_U
    md.project_basis(V)
    truth = p1[:, :5] @ rng.standard_normal(5)
    md.assemble = lambda mu, t, entries=None: np.array([truth[list(zip(rows, cols)).index(e)] for e in entries])
    data["interp_truth"] = truth
    data["interp_dofs_rc"] = np.array(md.dofs, dtype=np.int64)
    data["interp_PT_U"] = md.PT_U
    data["interp_fom"] = md._interpolate(mu={}, t=0.0, which=md.FOM)
    data["interp_rom"] = md.interpolate(mu={}, t=0.0, which=md.ROM)
    data["interp_thetas"] = md.compute_thetas(np.array([truth[list(zip(rows, cols)).index(e)] for e in md.dofs]))
    dd.load_fom_basis(basis=bases["random_orth_300x16"].copy())
    dd.project_basis(V2)
    tv = bases["random_orth_300x16"] @ rng.standard_normal(16)
    dd.assemble = lambda mu, t, entries=None: np.array([tv[i] for (i,) in entries])
    data["interp_vec_truth"] = tv
    data["interp_vec_fom"] = dd._interpolate(mu={}, t=0.0, which=dd.FOM)
    data["interp_vec_rom"] = dd._interpolate(mu={}, t=0.0, which=dd.ROM)

    # --- project_csr / eliminate_zeros (utils.py:96-113,152-168) ----------------
    A = fom.assemble_stiffness(dict(alpha_0=0.7, delta=0.3), 2.0)
    data["csr_indptr"] = A.indptr.astype(np.int64)
    data["csr_indices"] = A.indices.astype(np.int64)
    data["csr_data"] = A.data
    data["project_csr"] = ref.utils.project_csr(A, V)
    B = A.copy()
    B.data[::7] = 5e-16
    B.data[3::11] = -1e-15
    data["ez_data_in"] = B.data.copy()
    B2 = ref.utils.eliminate_zeros(B.copy())
    data["ez_indptr"] = B2.indptr.astype(np.int64)
    data["ez_indices"] = B2.indices.astype(np.int64)
    data["ez_data"] = B2.data

    # --- error metrics (rom/base.py:52-73, utils.py:173-212) -------------------
    u, ue = rng.standard_normal(37), rng.standard_normal(37)
    data["err_u"], data["err_ue"] = u, ue
    data["err"] = np.array(ref.base.Reductor._compute_error(u, ue))
    uN, uNs, Vs = rng.standard_normal(4), rng.standard_normal(7), rng.standard_normal((10, 7))
    data["diff_uN"], data["diff_uNs"], data["diff_Vs"] = uN, uNs, Vs
    data["diff"] = np.array(ref.utils.compute_rom_difference(uN, uNs, Vs))
    data["versions"] = np.array(repr(_versions()))
    np.savez_compressed(os.path.join(out, "deim.npz"), **data)
    print("deim.npz:", list(bases))


class RefFomAdapter:
    """Presents a MockBurgers to the reference's RomConstructorNonlinear.solve."""

    probe_location = []
    RUNTIME_PROCESS = False
    exact_solution = None

    def __init__(self, ref, fom):
        self._ref, self._fom = ref, fom
        self.BDF_SCHEME = fom.BDF_SCHEME
        self.domain = fom.domain
        self.dt = fom.dt
        self.V = None
        self.L = None
        self.is_setup = True
        self._cur = None

    def move_mesh(self, mu=None, t=None, back=False):
        self._cur = None if back else (mu, t)

    @property
    def x(self):
        mu, t = self._cur
        return self._fom.x_at(mu, t).reshape(-1, 1)

    def create_lifting_operator(self, mu, t, L):
        return (mu, t), None, None

    def interpolate_func(self, g, V, mu, t):
        arr = self._fom.lifting(mu, t)
        vec = types.SimpleNamespace(array=arr)
        return types.SimpleNamespace(vector=lambda: types.SimpleNamespace(vec=lambda: vec))

    def _mat(self, A):
        return self._ref.Matrix(A)

    def assemble_mass(self, mu, t):
        return self._mat(self._fom.assemble_mass(mu, t))

    def assemble_stiffness(self, mu, t):
        return self._mat(self._fom.assemble_stiffness(mu, t))

    def assemble_convection(self, mu, t):
        return self._mat(self._fom.assemble_convection(mu, t))

    def assemble_trilinear(self, mu, t, u_n):
        return self._mat(self._fom.assemble_trilinear(mu, t, u_n))

    def assemble_nonlinear_lifting(self, mu, t):
        return self._mat(self._fom.assemble_nonlinear_lifting(mu, t))

    def assemble_lifting(self, mu, t):
        return self._ref.Vector(self._fom.assemble_lifting(mu, t))


def gen_rom(ref, out):
    from romtime_amd.testing.mock import MockBurgers

    rng = np.random.RandomState(20260106)
    data = {}
    cases = []
    for r in (10, 24):
        for bdf2 in (False, True):
            fom = MockBurgers(domain=dict(L0=1.0, nx=120, T=0.5, nt=50),
                              Lt=lambda t, **mu: 1.0 - 0.1 * np.sin(mu["omega"] * t), bdf2=bdf2)
            fom.setup()
            x = np.linspace(0, 1, fom.Nh)
            modes = np.array([np.sin((k + 1) * np.pi * x) for k in range(r)]).T
            modes += 1e-3 * rng.standard_normal(modes.shape)
            modes[0, :] = modes[-1, :] = 0.0
            V, _ = np.linalg.qr(modes)
            mu = dict(alpha_0=0.05, delta=0.3, omega=9.0)
            rom = ref.rom.RomConstructorNonlinear(fom=RefFomAdapter(ref, fom), grid=None, name="golden")
            rom.setup(rnd=0)
            rom.basis = V
            rom.solve(mu=mu, step="online")
            key = f"r{r}_bdf{2 if bdf2 else 1}"
            cases.append(key)
            data[f"V__{key}"] = V
            data[f"rom__{key}"] = rom.solutions.rom
            data[f"fom__{key}"] = rom.solutions.fom
            data[f"ts__{key}"] = np.array(rom.solutions.ts)
    data["cases"] = np.array(cases)
    data["mu"] = np.array([0.05, 0.3, 9.0])  # alpha_0, delta, omega
    data["versions"] = np.array(repr(_versions()))
    np.savez_compressed(os.path.join(out, "rom.npz"), **data)
    print("rom.npz:", cases)


def gen_sampler(ref, out):
    from sklearn.model_selection import ParameterSampler
    import sklearn

    sys.path.insert(0, REF)
    from romtime.parameters import get_uniform_dist

    grid = {"delta": get_uniform_dist(min=0.01, max=2.0), "beta": get_uniform_dist(min=1.0, max=10.0),
            "alpha_0": get_uniform_dist(min=0.01, max=2.0)}  # tests/test_mdeim.py:43-47
    red = ref.base.Reductor(grid=grid)
    red.setup(rnd=np.random.RandomState(0))
    draws = list(red.build_sampling_space(num=8, rnd=np.random.RandomState(0)))
    keys = sorted(draws[0])
    arr = np.array([[d[k] for k in keys] for d in draws])
    np.savez_compressed(os.path.join(out, "sampler.npz"), keys=np.array(keys), draws=arr,
                        versions=np.array(repr(dict(sklearn=sklearn.__version__, **_versions()))))
    print("sampler.npz:", arr.shape)


# ---------------------------------------------------------------------------
# tree walks, truncation, reduced-basis construction (a2 / a14) -- the reference's own
# DiscreteEmpiricalInterpolation.run (deim.py:175-215,279-397), N-MDEIM run / truncate
# (nonlinear.py:49-104,159-212,405-468), RomConstructor.build_reduced_basis / truncate
# (rom.py:169-198,276-412) on the closed-form P1 FOM of romtime_amd.testing.mock
# ---------------------------------------------------------------------------
from romtime_amd.testing.walk_inputs import (PISTON_MUS, PISTON_SROM_TRUNCATE, PISTON_TOL_MU, PISTON_TOL_TIME,  # noqa: E402
                                              PISTON_TS, RB_CASES, RB_MUS,
                                              WALK_MUS, WALK_TS, nmdeim_states, piston_grid, rb_fom,
                                              walk_rich_operator, walk_solver, walk_state_operator)


def _as_dolfin(ref, fn):
    """The reference type-switches on dolfin Matrix (bilinear_to_csr, utils.py:76-93): full assemblies are wrapped
    in the Matrix stand-in, entry-wise assemblies stay plain arrays."""

    def assemble(**kw):
        out = fn(**kw)
        return out if kw.get("entries") is not None else ref.Matrix(out)

    return assemble


def _store_report(data, key, red, Stage):
    off = red.report[Stage.OFFLINE]
    n_mu = len(red.mu_space[Stage.OFFLINE])
    data[f"{key}__basis_time"] = np.array([off["basis-shape-time"][i] for i in range(n_mu)])
    data[f"{key}__after_walk"] = np.array(off["basis-shape-after-tree-walk"])
    data[f"{key}__final"] = np.array(off["basis-shape-final"])
    data[f"{key}__spectrum_mu"] = np.asarray(off["spectrum-mu"])
    data[f"{key}__energy_mu"] = np.asarray(off["energy-mu"])
    for i in range(n_mu):
        data[f"{key}__spectrum_time_{i}"] = np.asarray(off["spectrum-time"][i])


def _check_well_posed(name, report, tol_time=None, tol_mu=None, prefix=""):
    """A fixture must not sit on a knife edge of the truncation rules (pod.py:46-57): no singular value within 4 % of
    DROP_TOLERANCE where the drop branch decides, no energy within 10 % of (1 - tol) of ``tol`` where ``tol`` does."""
    def check(tag, s, e, tol):
        s, e = np.asarray(s), np.asarray(e)
        if tol:
            assert not np.any(np.abs(e - tol) < 0.1 * (1.0 - tol)), (name, tag, "energy on the tolerance", e, tol)
        else:
            assert not np.any((s > 0.96e-7) & (s < 1.04e-7)), (name, tag, "sigma on the drop tolerance", s)

    for i, s in report[prefix + "spectrum-time"].items():
        check(f"time {i}", s, report[prefix + "energy-time"][i], tol_time)
    check("mu", report[prefix + "spectrum-mu"], report[prefix + "energy-mu"], tol_mu)


def gen_walks(ref, out):
    Stage = ref.conv.Stage
    DEIM = ref.deim.DiscreteEmpiricalInterpolation
    MDEIM = ref.deim.MatrixDiscreteEmpiricalInterpolation
    NMDEIM = ref.nonlinear.MatrixDiscreteEmpiricalInterpolationNonlinear
    data = {}
    fom = walk_solver()
    params = {"ts": WALK_TS, "num_snapshots": len(WALK_MUS)}
    data["mus"] = np.array([[m["alpha_0"], m["beta"], m["delta"]] for m in WALK_MUS])
    data["ts"] = WALK_TS

    # -- DEIM.run on a vector functional, default truncation (drop branch at both levels) and num / tol variants
    for key, extra in (("deim_default", {}), ("deim_num", {"num_mu": 6, "num_time": 4}),
                       ("deim_tol", {"tol_mu": 1.0 - 1e-9, "tol_time": 1.0 - 1e-10})):
        d = DEIM(assemble=fom.assemble_forcing, grid=None, tree_walk_params=dict(params, **extra), name=key)
        d.setup(rnd=np.random.RandomState(0))
        d.run(mu_space=[dict(m) for m in WALK_MUS])
        data[f"{key}__basis_fom"] = d.basis_fom
        data[f"{key}__sigmas"] = d.sigmas
        data[f"{key}__dofs"] = np.array([i for (i,) in d.dofs], dtype=np.int64)
        data[f"{key}__PT_U"] = d.PT_U
        _store_report(data, key, d, Stage)
        data[f"{key}__margin"] = greedy_margins(d.basis_fom, data[f"{key}__dofs"])
        if not extra.get("num_mu"):
            _check_well_posed(key, d.report[Stage.OFFLINE], extra.get("tol_time"), extra.get("tol_mu"))

    # -- MDEIM.run: the reference's own acceptance operators (separable -> degenerate spectra) and a rich one
    ops_m = {"mdeim_stiffness": fom.assemble_stiffness, "mdeim_rich": walk_rich_operator(fom)}
    for key, fn in ops_m.items():
        md = MDEIM(assemble=_as_dolfin(ref, fn), grid=None, tree_walk_params=dict(params), name=key)
        ref.base.Reductor.setup(md, rnd=np.random.RandomState(0))     # MDEIM.setup samples the grid for a topology mu
        md.rows, md.cols = md.get_matrix_topology(mu=WALK_MUS[0], t=1.0)
        md.run(mu_space=[dict(m) for m in WALK_MUS])
        data[f"{key}__rows"] = np.array(md.rows, dtype=np.int64)
        data[f"{key}__cols"] = np.array(md.cols, dtype=np.int64)
        data[f"{key}__basis_fom"] = md.basis_fom
        data[f"{key}__sigmas"] = md.sigmas
        data[f"{key}__dofs"] = np.array(md.dofs, dtype=np.int64)
        data[f"{key}__PT_U"] = md.PT_U
        _store_report(data, key, md, Stage)
        _check_well_posed(key, md.report[Stage.OFFLINE])

    # -- N-MDEIM.run (three levels, all normalised) + truncate(n)
    x = np.linspace(0.0, 1.0, fom.Nh)
    psi = nmdeim_states(fom.Nh)
    nm = NMDEIM(assemble=_as_dolfin(ref, walk_state_operator(fom)), grid=None,
                tree_walk_params={"ts": WALK_TS[::2], "num_snapshots": 3}, name="nmdeim")
    ref.base.Reductor.setup(nm, rnd=np.random.RandomState(0))          # N-MDEIM.setup needs a FEniCS space (nonlinear.py:133-157)
    nm.rows, nm.cols = nm.get_matrix_topology(mu=WALK_MUS[0], t=1.0, u_n=x)
    nm.run(u_n=psi, mu_space=[dict(m) for m in WALK_MUS[:3]])
    key = "nmdeim"
    data[f"{key}__psi"] = psi
    data[f"{key}__rows"] = np.array(nm.rows, dtype=np.int64)
    data[f"{key}__cols"] = np.array(nm.cols, dtype=np.int64)
    data[f"{key}__basis_fom"] = nm.basis_fom
    data[f"{key}__sigmas"] = nm.sigmas
    data[f"{key}__dofs"] = np.array(nm.dofs, dtype=np.int64)
    data[f"{key}__PT_U"] = nm.PT_U
    _store_report(data, key, nm, Stage)
    _check_well_posed(key, nm.report[Stage.OFFLINE])
    n_cut = 3
    tr = nm.truncate(n_cut)
    data[f"{key}__trunc_n"] = np.array(n_cut)
    data[f"{key}__trunc_name"] = np.array(tr.name)
    data[f"{key}__trunc_basis_fom"] = tr.basis_fom
    data[f"{key}__trunc_dofs"] = np.array(tr.dofs, dtype=np.int64)
    data[f"{key}__trunc_PT_U"] = tr.PT_U
    data[f"{key}__trunc_final"] = np.array(tr.report[Stage.OFFLINE]["basis-shape-final"])
    # interpolation (nonlinear.py:247-283, FOM form) at a training (mu, t) of a state inside the training span whose
    # coefficients sum to one (the operator is affine in the state): the interpolant reproduces the operator
    u_probe = psi @ np.array([0.4, -0.2, 0.6, 0.2])
    data[f"{key}__probe_u"] = u_probe
    data[f"{key}__probe_t"] = np.array(WALK_TS[2])
    data[f"{key}__interp_fom"] = nm._interpolate(mu=WALK_MUS[1], t=WALK_TS[2], u_n=u_probe, which=nm.FOM)

    # -- RomConstructorNonlinear.build_reduced_basis (two-level POD of FOM solves) + truncate
    for key, tolerances, num_basis in RB_CASES:
        bfom = rb_fom()
        mus = RB_MUS
        ad = RefFomAdapter(ref, bfom)
        ad.setup = lambda: None
        ad.update_parametrization = bfom.update_parametrization

        def solve(ad=ad, bfom=bfom):
            bfom.solve()
            ad.solutions = bfom.solutions
            ad.nonlinear_snapshots = bfom.nonlinear_snapshots

        ad.solve = solve
        rom = ref.rom.RomConstructorNonlinear(fom=ad, grid=None, name="S-ROM")
        rom.setup(rnd=0)
        sols = rom.build_reduced_basis(mu_space=[dict(m) for m in mus], num_basis=num_basis, tolerances=tolerances)
        off = rom.report[Stage.OFFLINE]
        data[f"{key}__mus"] = np.array([[m["alpha_0"], m["delta"], m["omega"]] for m in mus])
        data[f"{key}__basis"] = rom.basis
        data[f"{key}__basis_nonlinear"] = rom.basis_nonlinear
        data[f"{key}__fom_solution_1"] = sols[1]
        data[f"{key}__basis_time"] = np.array([off["basis-shape-time"][i] for i in range(3)])
        data[f"{key}__after_walk"] = np.array(off["basis-shape-after-tree-walk"])
        data[f"{key}__final"] = np.array(off["basis-shape-final"])
        data[f"{key}__spectrum_mu"] = np.asarray(off["spectrum-mu"])
        data[f"{key}__energy_mu"] = np.asarray(off["energy-mu"])
        data[f"{key}__N_basis_time"] = np.array([off["N-basis-shape-time"][i] for i in range(3)])
        data[f"{key}__N_after_walk"] = np.array(off["N-basis-shape-after-tree-walk"])
        data[f"{key}__N_final"] = np.array(off["N-basis-shape-final"])
        data[f"{key}__N_spectrum_mu"] = np.asarray(off["N-spectrum-mu"])
        for i in range(3):
            data[f"{key}__spectrum_time_{i}"] = np.asarray(off["spectrum-time"][i])
            data[f"{key}__N_spectrum_time_{i}"] = np.asarray(off["N-spectrum-time"][i])
        if key != "rb_num":
            _check_well_posed(key, off, tolerances.get("tol_time"), tolerances.get("tol_mu"))
            _check_well_posed(key + " (nonlinear)", off, tolerances.get("tol_time"), None, prefix="N-")
        tr = rom.truncate(2)
        data[f"{key}__trunc_basis"] = tr.basis
        data[f"{key}__trunc_final"] = np.array(tr.report[Stage.OFFLINE]["basis-shape-final"])
        print(f"  {key}: time {data[f'{key}__basis_time']} -> {int(data[f'{key}__after_walk'])} -> {rom.N};"
              f" nonlinear -> {rom.basis_nonlinear.shape[1]}")
    data["versions"] = np.array(repr(_versions()))
    np.savez_compressed(os.path.join(out, "walks.npz"), **data)
    print("walks.npz:", len(data), "arrays")


# ---------------------------------------------------------------------------
# f1: the real caller.  HyperReducedPiston (rom/hrom.py:979-1182) and its base class's run_offline_rom,
# run_offline_hyperreduction, project_reductors, evaluate_validation / evaluate_online -> _evaluate
# (rom/hrom.py:308-342, 419-452, 459-626) drive an S-ROM / ROM pair of RomConstructorNonlinear with all six operators
# hyper-reduced.  Only what needs FEniCS is stepped around: ``setup`` (builds the FEniCS FOM; the duck-typed mock
# is attached instead) and the two ``N-MDEIM.setup(rnd, V)`` calls inside ``setup_hyperreduction`` (they interpolate
# an Expression on the FE space to get a topology; the topology is read off the mock with u_n = x).
# ---------------------------------------------------------------------------
class PistonFomAdapter(RefFomAdapter):
    """RefFomAdapter + what the driver itself touches (hrom.py:504-626)."""

    def __init__(self, ref, fom):
        super().__init__(ref, fom)
        self.V = None

    def setup(self):
        pass

    def update_parametrization(self, mu):
        self._fom.update_parametrization(mu)

    def solve(self):
        self._fom.solve()
        self.solutions = self._fom.solutions
        self.nonlinear_snapshots = self._fom.nonlinear_snapshots

    def _wrap(self, fn, **kw):
        out = fn(**kw)
        return out if kw.get("entries") is not None else self._mat(out)

    def assemble_mass(self, mu=None, t=None, entries=None):
        return self._wrap(self._fom.assemble_mass, mu=mu, t=t, entries=entries)

    def assemble_stiffness(self, mu=None, t=None, entries=None):
        return self._wrap(self._fom.assemble_stiffness, mu=mu, t=t, entries=entries)

    def assemble_convection(self, mu=None, t=None, entries=None):
        return self._wrap(self._fom.assemble_convection, mu=mu, t=t, entries=entries)

    def assemble_nonlinear_lifting(self, mu=None, t=None, entries=None):
        return self._wrap(self._fom.assemble_nonlinear_lifting, mu=mu, t=t, entries=entries)

    def assemble_trilinear(self, mu=None, t=None, u_n=None, entries=None):
        return self._wrap(self._fom.assemble_trilinear, mu=mu, t=t, u_n=u_n, entries=entries)

    def assemble_lifting(self, mu=None, t=None, entries=None):
        out = self._fom.assemble_lifting(mu, t, entries=entries)
        return out if entries is not None else self._ref.Vector(out)

    assemble_rhs = assemble_lifting       # Burgers has no forcing: the right-hand side functional is the lifting term

    def assemble_nonlinear(self, mu=None, t=None, u_n=None, entries=None):
        raise NotImplementedError         # the NONLINEAR model is switched off (hrom.py:1112-1140 only runs TRILINEAR)

    def compute_mass_conservation(self, mu, ts, solutions, which):
        """FEniCS quadrature in the reference (fom/nonlinear.py:627-700): reporting only, not on the path."""
        return {"which": which, "timesteps": list(ts), "mass": [float(np.sum(u)) for u in solutions]}


def run_piston_driver(ref, classes=None):
    """Drive HyperReducedPiston's own methods (the reference's code) over ``classes`` = dict(rom=, deim=, mdeim=,
    nmdeim=, reductor=): the reference's classes by default (fixture generation), or romtime_amd's - the drop-in
    check of tests/test_hrom_flow.py, where the names the driver module bound at import are re-pointed first."""
    import tempfile

    RP, OT, Stage = ref.conv.RomParameters, ref.conv.OperatorType, ref.conv.Stage
    hrom = ref.hrom
    if classes is None:
        classes = dict(rom=ref.rom.RomConstructorNonlinear, deim=ref.deim.DiscreteEmpiricalInterpolation,
                       mdeim=ref.deim.MatrixDiscreteEmpiricalInterpolation,
                       nmdeim=ref.nonlinear.MatrixDiscreteEmpiricalInterpolationNonlinear, reductor=ref.base.Reductor,
                       compute_rom_difference=ref.utils.compute_rom_difference)
    saved = {k: getattr(hrom, k) for k in ("DiscreteEmpiricalInterpolation", "MatrixDiscreteEmpiricalInterpolation",
                                           "MatrixDiscreteEmpiricalInterpolationNonlinear", "RomConstructorNonlinear",
                                           "compute_rom_difference")}
    hrom.DiscreteEmpiricalInterpolation = classes["deim"]
    hrom.MatrixDiscreteEmpiricalInterpolation = classes["mdeim"]
    hrom.MatrixDiscreteEmpiricalInterpolationNonlinear = classes["nmdeim"]
    hrom.RomConstructorNonlinear = classes["rom"]
    hrom.compute_rom_difference = classes["compute_rom_difference"]
    try:
        def make_driver():
            grid = piston_grid()
            rnd = np.random.RandomState(0)
            walk = {"ts": PISTON_TS, RP.NUM_SNAPSHOTS: None}
            H = hrom.HyperReducedPiston(
                grid=grid, fom_params={}, rom_params={RP.NUM_SNAPSHOTS: None, RP.SROM_TRUNCATE: PISTON_SROM_TRUNCATE,
                                                      RP.TOL_TIME: PISTON_TOL_TIME, RP.TOL_MU: PISTON_TOL_MU,
                                                      RP.SROM_KEEP: None, RP.NMDEIM_SIZE: None},
                deim_params=dict(walk), mdeim_params=dict(walk), mdeim_nonlinear_params=dict(walk),
                models={OT.MASS: True, OT.STIFFNESS: True, OT.RHS: True, OT.CONVECTION: True, OT.NONLINEAR_LIFTING: True,
                        OT.TRILINEAR: True}, rnd=rnd)
            fom = PistonFomAdapter(ref, rb_fom())
            # -- HyperReducedPiston.setup (hrom.py:1003-1038) without the FEniCS solver
            H.fom = fom
            H.rom = classes["rom"](fom=fom, grid=grid, name="ROM")
            H.rom.setup(rnd=rnd)
            H.srom = classes["rom"](fom=fom, grid=grid, name="S-ROM")
            H.srom.setup(rnd=rnd)
            # -- setup_hyperreduction (hrom.py:274-306 + 1040-1086): same constructors, same names, same callbacks
            hrom.HyperReducedOrderModelFixed.setup_hyperreduction(H)          # RHS, Mass, Stiffness: runs unmodified
            MDEIM, NMDEIM = classes["mdeim"], classes["nmdeim"]
            H.mdeim_convection = MDEIM(name=OT.CONVECTION, assemble=fom.assemble_convection, grid=grid,
                                       tree_walk_params=H.mdeim_params)
            H.mdeim_trilinear_lifting = MDEIM(name=OT.NONLINEAR_LIFTING, assemble=fom.assemble_nonlinear_lifting, grid=grid,
                                              tree_walk_params=H.mdeim_params)
            H.mdeim_trilinear = NMDEIM(name=OT.TRILINEAR, assemble=fom.assemble_trilinear, grid=grid,
                                       tree_walk_params=H.mdeim_nonlinear_params)
            H.mdeim_convection.setup(rnd=rnd)
            H.mdeim_trilinear_lifting.setup(rnd=rnd)
            classes["reductor"].setup(H.mdeim_trilinear, rnd=rnd)
            x = np.linspace(0.0, 1.0, fom._fom.Nh)
            H.mdeim_trilinear.rows, H.mdeim_trilinear.cols = H.mdeim_trilinear.get_matrix_topology(mu=PISTON_MUS[0], t=1.0,
                                                                                                 u_n=x)
            return H

        H = make_driver()
        captured = {}

        def capture(rom_obj, label):
            orig = rom_obj.solve

            def solve(mu, step):
                idx = orig(mu=mu, step=step)
                captured[(label, step, idx)] = (rom_obj.solutions.rom.copy(), rom_obj.solutions.fom.copy())
                return idx

            rom_obj.solve = solve

        here = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                mus = [dict(m) for m in PISTON_MUS]
                H.run_offline_rom(mu_space=mus)                                   # hrom.py:308-342
                H.run_offline_hyperreduction(mu_space=mus, evaluate=False)        # hrom.py:1088-1140
                H.project_reductors()                                             # hrom.py:265-272
                capture(H.rom, "rom")
                capture(H.srom, "srom")
                H.evaluate_validation()                                           # hrom.py:476-481 -> _evaluate
                H.evaluate_online(params={"num": 2}, rnd=np.random.RandomState(1))  # hrom.py:483-502
                # -- f2: leave the artefacts behind and resume from them in a fresh driver (hrom.py:137-177, 344-417)
                H.dump_mu_space()
                H.dump_reduced_basis()
                H.dump_nonlinear_basis()
                H.dump_validation_fom()
                written = sorted(os.listdir(tmp))
                H2 = make_driver()
                H2.start_from_existing_basis()
                H2.project_reductors()
                H2.evaluate_validation()
            finally:
                os.chdir(here)
    finally:
        for k, v in saved.items():
            setattr(hrom, k, v)
    data = {}
    data["offline_mus"] = np.array([[m[k] for k in ("a0", "omega", "delta", "alpha_0")] for m in PISTON_MUS])
    online = H.rom.mu_space[Stage.ONLINE]
    data["online_mus"] = np.array([[m[k] for k in ("a0", "omega", "delta", "alpha_0")] for m in online])
    data["online_mach"] = np.array([m["piston_mach"] for m in online])
    data["N_rom"], data["N_srom"] = np.array(H.rom.N), np.array(H.srom.N)
    for name, red in (("rhs", H.deim_rhs), ("mass", H.mdeim_mass), ("stiffness", H.mdeim_stiffness),
                      ("convection", H.mdeim_convection), ("nonlinear_lifting", H.mdeim_trilinear_lifting),
                      ("trilinear", H.mdeim_trilinear)):
        data[f"N__{name}"] = np.array(red.N)
        data[f"dofs__{name}"] = np.array(red.dofs, dtype=np.int64)
        # equal singular values among the kept modes: the basis columns (hence the entries) are LAPACK's arbitrary pick
        sig = np.asarray(red.sigmas if red.sigmas is not None else H.srom.report[Stage.OFFLINE]["N-spectrum-mu"])
        kept = sig[: red.N + 1] if red.N < len(sig) else sig[: red.N]
        data[f"cluster__{name}"] = np.array(bool(np.any(np.abs(np.diff(kept)) < 1e-6 * sig[0])))
    data["srom_basis"] = H.srom.basis
    data["srom_basis_nonlinear"] = H.srom.basis_nonlinear
    off = H.srom.report[Stage.OFFLINE]
    data["srom_spectrum_mu"] = np.asarray(off["spectrum-mu"])
    data["srom_N_spectrum_mu"] = np.asarray(off["N-spectrum-mu"])
    for i in range(3):
        data[f"srom_spectrum_time_{i}"] = np.asarray(off["spectrum-time"][i])
        data[f"srom_N_spectrum_time_{i}"] = np.asarray(off["N-spectrum-time"][i])
        data[f"srom_basis_time_{i}"] = np.array(off["basis-shape-time"][i])
    for which in (Stage.VALIDATION, Stage.ONLINE):
        for idx, payload in H.errors[which].items():
            for kind, arr in payload.items():
                data[f"errors__{which}__{idx}__{kind}"] = np.asarray(arr)
            for label in ("rom", "srom"):
                uN, uh = captured[(label, which, idx)]
                data[f"{label}_uN__{which}__{idx}"] = uN
                data[f"{label}_uh__{which}__{idx}"] = uh
    data["validation_solution_1"] = H.validation_solutions[1]
    data["files_written"] = np.array(written)
    data["resume_N_rom"], data["resume_N_srom"] = np.array(H2.rom.N), np.array(H2.srom.N)
    data["resume_offline_mus"] = np.array([[m[k] for k in ("a0", "omega", "delta", "alpha_0")] for m in H2.rom.mu_space[Stage.OFFLINE]])
    for idx, payload in H2.errors[Stage.VALIDATION].items():
        for kind, arr in payload.items():
            data[f"resume_errors__{idx}__{kind}"] = np.asarray(arr)
    return data, H


def gen_hrom(ref, out):
    Stage = ref.conv.Stage
    data, H = run_piston_driver(ref)
    data["versions"] = np.array(repr(_versions()))
    np.savez_compressed(os.path.join(out, "hrom.npz"), **data)
    print("hrom.npz: N_rom", H.rom.N, "N_srom", H.srom.N, {k[3:]: int(v) for k, v in data.items() if k.startswith("N__")},
          "online mach", data["online_mach"])
    for which in (Stage.VALIDATION, Stage.ONLINE):
        for idx, payload in H.errors[which].items():
            print("  ", which, idx, {k: float(np.max(v)) for k, v in payload.items()})


class HeatFomAdapter(RefFomAdapter):
    """Presents the closed-form MFP1 heat problem (romtime_amd.testing.mock.MockHeatEquation) to the reference's linear
    ROM classes: dolfin-like Matrix / Vector stand-ins around its operators and load vectors."""

    def assemble_forcing(self, mu, t):
        return self._ref.Vector(self._fom.assemble_forcing(mu, t))


def gen_heat(ref, out):
    """The reference's LINEAR ROM classes (rom.py:34-736) on the heat problem of config 1, fixed and moving mesh.
    At v0 their `solve` cannot run (it calls assemble_system / assemble_system_rhs with the argument lists of the
    nonlinear class, rom.py:487-488 vs :557-573, :714-736: TypeError), so the fixture pins what CAN run - every reduced
    operator and vector, and the reference's own K_N and b_N formulas - at several (mu, t)."""
    from romtime_amd.testing.mock import MockHeatEquation
    from romtime_amd.testing.walk_inputs import heat_problem

    rng = np.random.RandomState(20260110)
    data = {}
    cases = []
    for moving in (False, True):
        fom, V, states = heat_problem(moving)
        ad = HeatFomAdapter(ref, fom)
        cls = ref.rom.RomConstructorMoving if moving else ref.rom.RomConstructor
        rom = cls(fom=ad, grid=None, name="golden-heat")
        rom.setup(rnd=0)
        rom.basis = V
        key = "moving" if moving else "fixed"
        cases.append(key)
        data[f"V__{key}"] = V
        for q, (mu, t) in enumerate(states):
            uN = rng.standard_normal(V.shape[1])
            MN, KN = rom.assemble_system(mu, t)                       # rom.py:565-573 / :714-736
            bN = rom.assemble_system_rhs(mu, t, uN, MN)               # rom.py:557-563 (argument order of the base class)
            tag = f"{key}_{q}"
            data[f"uN__{tag}"] = uN
            data[f"MN__{tag}"], data[f"KN__{tag}"], data[f"bN__{tag}"] = MN, KN, bN
            data[f"AN__{tag}"] = rom.assemble_stiffness(mu, t)
            data[f"fN__{tag}"] = rom.assemble_forcing(mu, t)
            data[f"fgN__{tag}"] = rom.assemble_lifting(mu, t)
            data[f"rhsN__{tag}"] = rom.assemble_rhs(mu, t)
            if moving:
                data[f"CN__{tag}"] = rom.assemble_convection(mu, t)
        data[f"n_states__{key}"] = np.array(len(states))
    data["cases"] = np.array(cases)
    data["versions"] = np.array(repr(_versions()))
    np.savez_compressed(os.path.join(out, "heat.npz"), **data)
    print("heat.npz:", cases, {k: int(data[f"n_states__{k}"]) for k in cases})


def main():
    if not os.path.isdir(REF):
        print("reference not present; nothing to do")
        return
    ref = load_reference()
    gen_orth(ref, HERE)
    gen_deim(ref, HERE)
    gen_rom(ref, HERE)
    gen_sampler(ref, HERE)
    gen_walks(ref, HERE)
    gen_hrom(ref, HERE)
    gen_heat(ref, HERE)


if __name__ == "__main__":
    main()
