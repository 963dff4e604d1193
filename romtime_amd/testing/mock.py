"""Closed-form 1-D P1 full-order models (NumPy/SciPy, no FEniCS).

These stand where the reference's FEniCS solvers stand in its tests
(``src/romtime/testing/mock.py:6-144`` for the linear operators,
``src/romtime/fom/nonlinear.py:322-494`` for the Burgers/piston forms): they
serve the ``assemble_*(mu, t[, entries][, u_n])`` callbacks the (M)DEIM and ROM
classes call, returning ``scipy.sparse.csr_matrix`` / ``numpy.ndarray`` instead
of dolfin objects.  Element matrices are the exact P1 integrals on a uniform
interval mesh of ``nx`` cells scaled to ``[0, L(t)]``:

    mass        h/6 [[2, 1], [1, 2]]
    stiffness   alpha/h [[1, -1], [-1, 1]],   alpha = alpha_0 (1 + t^2)   (mock.py:30-48)
    convection  -int u' v  = -1/2 [[-1, 1], [-1, 1]]                        (mock.py:69-85)
    trilinear   int w u' v,  w = u_n linear on the cell                     (fom/nonlinear.py:398-418)

Homogeneous Dirichlet rows (first and last dof) are replaced by identity rows,
as ``DirichletBC.apply`` does (``fom/base.py:501-521``); local (entry-wise)
assembly returns 1.0 / 0.0 on those (``fom/base.py:536-546``).
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix


class MockSolver:
    """Linear 1-D P1 operators on a (possibly moving) interval."""

    DIRICHLET_ENTRY = 1.0  # fom/base.py:49
    DIRICHLET_VALUE = 0.0  # fom/base.py:50
    BDF_SCHEME = "2"
    RUNTIME_PROCESS = False

    def __init__(self, domain, forcing_term=None, Lt=None):
        self.domain = dict(domain)
        self.forcing_term = forcing_term  # callable f(x, t, **mu) or None
        self.Lt = Lt  # callable L(t, **mu) -> scale factor, or None (fixed mesh)
        self.is_setup = False
        self.exact_solution = None
        self.mu = None

    # -- bookkeeping the ROM classes touch ---------------------------------
    def setup(self):
        self.nx = int(self.domain["nx"])
        self.Nh = self.nx + 1
        self.is_setup = True

    def update_parametrization(self, new):
        self.mu = dict(new)

    @property
    def dt(self):
        return self.domain["T"] / self.domain["nt"]

    def _L(self, mu, t):
        L0 = self.domain["L0"]
        return L0 * (self.Lt(t=t, **mu) if self.Lt is not None else 1.0)

    def x_at(self, mu, t):
        return np.linspace(0.0, self._L(mu, t), self.Nh)

    # -- element-level assembly ---------------------------------------------
    def _assemble_matrix(self, elem, entries):
        """elem: (nx, 2, 2) element matrices.  Global CSR or values at entries."""
        nx, Nh = self.nx, self.Nh
        if entries is not None:
            out = np.empty(len(entries))
            for k, (i, j) in enumerate(entries):
                if i in (0, Nh - 1):
                    out[k] = self.DIRICHLET_ENTRY if i == j else 0.0
                    continue
                v = 0.0
                for e in (i - 1, i):  # cells touching dof i
                    if 0 <= e < nx and e <= j <= e + 1:
                        v += elem[e, i - e, j - e]
                out[k] = v
            return out
        e = np.arange(nx)
        rows = np.concatenate([e, e, e + 1, e + 1])
        cols = np.concatenate([e, e + 1, e, e + 1])
        vals = np.concatenate([elem[:, 0, 0], elem[:, 0, 1], elem[:, 1, 0], elem[:, 1, 1]])
        keep = (rows != 0) & (rows != Nh - 1)
        rows = np.concatenate([rows[keep], [0, Nh - 1]])
        cols = np.concatenate([cols[keep], [0, Nh - 1]])
        vals = np.concatenate([vals[keep], [1.0, 1.0]])
        A = csr_matrix((vals, (rows, cols)), shape=(Nh, Nh))
        A.sum_duplicates()
        A.sort_indices()
        return A

    def _assemble_vector(self, elem, entries):
        """elem: (nx, 2) element vectors."""
        nx, Nh = self.nx, self.Nh
        f = np.zeros(Nh)
        np.add.at(f, np.arange(nx), elem[:, 0])
        np.add.at(f, np.arange(nx) + 1, elem[:, 1])
        f[0] = f[-1] = self.DIRICHLET_VALUE
        if entries is not None:
            return np.array([f[i] for (i,) in entries])
        return f

    def _h(self, mu, t):
        return self._L(mu, t) / self.nx

    def assemble_mass(self, mu=None, t=None, entries=None):
        h = self._h(mu, t)
        elem = np.tile(h / 6.0 * np.array([[2.0, 1.0], [1.0, 2.0]]), (self.nx, 1, 1))
        return self._assemble_matrix(elem, entries)

    def assemble_stiffness(self, mu=None, t=None, entries=None):
        h = self._h(mu, t)
        alpha = mu["alpha_0"] * (1.0 + t * t)
        elem = np.tile(alpha / h * np.array([[1.0, -1.0], [-1.0, 1.0]]), (self.nx, 1, 1))
        return self._assemble_matrix(elem, entries)

    def assemble_convection(self, mu=None, t=None, entries=None):
        elem = np.tile(-0.5 * np.array([[-1.0, 1.0], [-1.0, 1.0]]), (self.nx, 1, 1))
        return self._assemble_matrix(elem, entries)

    def assemble_forcing(self, mu, t, entries=None):
        x = self.x_at(mu, t)
        h = self._h(mu, t)
        fx = self.forcing_term(x, t, **mu)
        fa, fb = fx[:-1], fx[1:]
        elem = np.stack([h / 6.0 * (2.0 * fa + fb), h / 6.0 * (fa + 2.0 * fb)], axis=1)
        return self._assemble_vector(elem, entries)

    def assemble_lifting(self, mu, t, entries=None):
        return np.zeros(self.Nh) if entries is None else np.zeros(len(entries))


class MockHeatEquation(MockSolver):
    """The manufactured heat problem MFP1 (problems/mfp1.py:19-75) on a fixed or moving interval, in closed form.

    ``mu`` keys: ``delta``, ``beta``, ``alpha_0`` (+ whatever ``Lt`` / ``dLt_dt`` take).  Dirichlet data
    ``b0 = 1 - exp(-beta t)``, ``bL = b0 (1 + delta^2 L^2)`` and their time derivatives (mfp1.py:27-35), forcing
    ``beta exp(-beta t)(1 + delta^2 x^2) - 2 delta^2 alpha_0 (1 - exp(-beta t))`` (mfp1.py:38-39), constant diffusivity
    ``alpha_0`` (fom/heat.py:42-55).  The reference hands these to FEniCS as ``Expression(..., degree=2)``
    (fom/heat.py:119, fom/base.py:452-495): each is interpolated at the vertices and the midpoint of every cell and the
    product with the P1 test function integrated exactly, i.e. per cell

        int f phi_l = h/6 (f_l + 2 f_m),     int f phi_r = h/6 (2 f_m + f_r)

    (Simpson's rule on a cubic) - exact for the quadratic forcing, which the P1-interpolated rule of
    ``MockSolver.assemble_forcing`` is not.  Pinned by the reference's known-answer tables ``expected_mat_fh`` /
    ``expected_mat_fgh_time`` (tests/test_mpf1.py:288-302; tests/test_oracle_golden.py)."""

    def __init__(self, domain, Lt=None, dLt_dt=None):
        super().__init__(domain=domain, forcing_term=None, Lt=Lt)
        self.dLt_dt = dLt_dt   # callable dL/dt (t, **mu) of the SCALE factor, or None (fom/base.py:407-412)
        self.BDF_SCHEME = "1"

    def assemble_stiffness(self, mu=None, t=None, entries=None):
        h = self._h(mu, t)
        elem = np.tile(mu["alpha_0"] / h * np.array([[1.0, -1.0], [-1.0, 1.0]]), (self.nx, 1, 1))
        return self._assemble_matrix(elem, entries)

    # -- the problem's data as polynomials in the physical coordinate ---------------------------------------------
    def forcing_poly(self, mu, t):
        """(a0, a1, a2) of the forcing a0 + a1 x + a2 x^2 (mfp1.py:38-39)."""
        e = np.exp(-mu["beta"] * t)
        d2 = mu["delta"] ** 2
        return np.array([mu["beta"] * e - 2.0 * d2 * mu["alpha_0"] * (1.0 - e), 0.0, mu["beta"] * e * d2])

    def boundary_data(self, mu, t):
        """b0, bL, grad_g and the coefficients (d0, d1) of dg_dt = d0 + d1 x (fom/base.py:377-495)."""
        L = self._L(mu, t)
        dL = self.domain["L0"] * self.dLt_dt(t=t, **mu) if self.dLt_dt is not None else 0.0
        e = np.exp(-mu["beta"] * t)
        d2 = mu["delta"] ** 2
        b0, bL = 1.0 - e, (1.0 - e) * (1.0 + d2 * L * L)
        db0 = mu["beta"] * e
        dbL = mu["beta"] * e * (1.0 + d2 * L * L) + 2.0 * (1.0 - e) * d2 * L * dL        # mfp1.py:35
        d1 = (dbL - db0) / L + (b0 - bL) * dL / (L * L)     # linear interpolation + moving-boundary effect, base.py:413-421
        return dict(b0=b0, bL=bL, grad_g=(bL - b0) / L, d0=db0, d1=d1)

    def lifting_poly(self, mu, t):
        """(a0, a1, 0) of dg_dt: the lifting vector is MINUS the degree-2 load of it at interior dofs."""
        bd = self.boundary_data(mu, t)
        return np.array([bd["d0"], bd["d1"], 0.0])

    def lifting(self, mu, t):
        bd = self.boundary_data(mu, t)
        x, L = self.x_at(mu, t), self._L(mu, t)
        return bd["bL"] * x / L + bd["b0"] * (L - x) / L

    def exact_solution_at(self, mu, t):
        x = self.x_at(mu, t)
        return (1.0 - np.exp(-mu["beta"] * t)) * (1.0 + mu["delta"] ** 2 * x * x)          # mfp1.py:45

    # -- degree-2 load vectors ---------------------------------------------------------------------------------------
    def _p2_elements(self, poly, mu, t):
        x, h = self.x_at(mu, t), self._h(mu, t)
        f = lambda z: poly[0] + poly[1] * z + poly[2] * z * z
        fl, fm, fr = f(x[:-1]), f(0.5 * (x[:-1] + x[1:])), f(x[1:])
        return np.stack([h / 6.0 * (fl + 2.0 * fm), h / 6.0 * (2.0 * fm + fr)], axis=1)

    def assemble_forcing(self, mu, t, entries=None):
        return self._assemble_vector(self._p2_elements(self.forcing_poly(mu, t), mu, t), entries)

    def assemble_lifting(self, mu, t, entries=None):
        """-(int dg_dt v + alpha grad_g . grad v), fom/heat.py:131-169."""
        bd = self.boundary_data(mu, t)
        elem = self._p2_elements(self.lifting_poly(mu, t), mu, t)
        flux = mu["alpha_0"] * bd["grad_g"]                 # int grad_g phi' over a cell: -/+ grad_g (cancels at interior dofs)
        elem = -(elem + np.array([-flux, flux])[None, :])
        return self._assemble_vector(elem, entries)

    def assemble_rhs(self, mu, t, entries=None):
        return self.assemble_forcing(mu, t, entries) + self.assemble_lifting(mu, t, entries)   # fom/heat.py:171-188

    def mesh_velocity_amplitude(self, mu, t):
        """w(L): the mesh velocity of the ALE form is the ramp w(x) = x dLt_dt / Lt (fom/heat.py:242-249)."""
        if self.dLt_dt is None:
            return 0.0
        return self._L(mu, t) * self.dLt_dt(t=t, **mu) / self.Lt(t=t, **mu)

    def assemble_convection(self, mu=None, t=None, entries=None):
        """The ALE term  -int w u' v  (fom/heat.py:267-285): w is linear, so its nodal interpolant is exact and the
        element matrices are those of the trilinear form with w as the state, negated.  Zero on a fixed mesh (where
        the reference's fixed-mesh solver has no such operator at all)."""
        w = self.mesh_velocity_amplitude(mu, t) * np.arange(self.Nh) / self.nx
        wa, wb = w[:-1], w[1:]
        a, b = (2.0 * wa + wb) / 6.0, (wa + 2.0 * wb) / 6.0
        elem = -np.stack([np.stack([-a, a], axis=1), np.stack([-b, b], axis=1)], axis=1)
        return self._assemble_matrix(elem, entries)

    def assemble_system(self, mu, t):
        """(M, K) of a BDF1 step, K = M + dt (C + A) (fom/heat.py:57-82 fixed, :251-262 moving)."""
        M = self.assemble_mass(mu, t)
        K = M + self.dt * (self.assemble_stiffness(mu, t) + self.assemble_convection(mu, t))
        return M, K

    def solve(self):
        """Full-order BDF1 time loop (host, SciPy sparse LU) producing ``solutions`` the way
        ``OneDimensionalSolver.solve`` does (fom/base.py:693-831): zero initial condition (u0 = 0 and g(0) = 0 for this
        problem), ``snapshots`` hold the homogeneous part, ``fom`` adds the lifting function on the moved mesh."""
        from scipy.sparse.linalg import spsolve

        from ..storage import SolutionsStorage

        mu, dt = self.mu, self.dt
        u = np.zeros(self.Nh)
        ts, homog, full, xs = [], [], [], []
        self.errors = []
        for step in range(int(self.domain["nt"])):
            t = (step + 1) * dt
            M, K = self.assemble_system(mu, t)
            rhs = M.dot(u) + dt * self.assemble_rhs(mu, t)                  # fom/heat.py:57-67
            rhs[0] = rhs[-1] = 0.0
            u = spsolve(K.tocsc(), rhs)
            ts.append(t)
            homog.append(u.copy())
            full.append(u + self.lifting(mu, t))
            xs.append(self.x_at(mu, t).reshape(-1, 1))
            self.errors.append(float(np.sqrt(np.mean((full[-1] - self.exact_solution_at(mu, t)) ** 2))))
        self.solutions = SolutionsStorage(ts=ts, mu=mu, domain=np.hstack(xs), fom=np.array(full).T,
                                          snapshots=np.array(homog).T)
        self.nonlinear_snapshots = None

    def p1_closed_form(self, mus, ts):
        """As MockSolver.p1_closed_form, plus what ``rt_p1_local_assembly("load_p2", poly=...)`` needs for the two load
        vectors: (nt, n_mu, 3) polynomial coefficients of the forcing and of dg_dt; and ``mesh_velocity`` (nt, n_mu): the
        amplitude of the ramp w of the ALE convection operator = minus the trilinear kind with that ramp."""
        out = _p1_closed_form(self, mus, ts)
        out["alpha"] = np.broadcast_to(np.array([mu["alpha_0"] for mu in mus])[None, :], out["h"].shape).copy()
        out["mesh_velocity"] = np.array([[self.mesh_velocity_amplitude(mu, t) for mu in mus] for t in np.asarray(ts, dtype=float)])
        out["forcing_poly"] = np.array([[self.forcing_poly(mu, t) for mu in mus] for t in np.asarray(ts, dtype=float)])
        out["lifting_poly"] = np.array([[self.lifting_poly(mu, t) for mu in mus] for t in np.asarray(ts, dtype=float)])
        return out


class MockBurgers(MockSolver):
    """Burgers/piston-type operators for the online loop (rom/rom.py:877-929).

    ``mu`` keys: ``alpha_0`` (viscosity), ``delta``, ``omega`` (lifting motion).
    """

    def __init__(self, domain, Lt=None, bdf2=True):
        super().__init__(domain=domain, forcing_term=None, Lt=Lt)
        self.bdf2 = bdf2
        self.BDF_SCHEME = "2" if bdf2 else "1"

    @property
    def nt(self):
        return int(self.domain["nt"])

    def _trilinear_elem(self, w):
        wa, wb = w[:-1], w[1:]
        a = (2.0 * wa + wb) / 6.0
        b = (wa + 2.0 * wb) / 6.0
        return np.stack([np.stack([-a, a], axis=1), np.stack([-b, b], axis=1)], axis=1)

    def assemble_trilinear(self, mu, t, u_n, entries=None):
        return self._assemble_matrix(self._trilinear_elem(np.asarray(u_n)), entries)

    def lifting(self, mu, t):
        """Nodal values g_h of the lifting function on the moved mesh."""
        x = self.x_at(mu, t)
        L = self._L(mu, t)
        amp = mu["delta"] * np.sin(mu["omega"] * t)
        return amp * x / L

    def assemble_nonlinear_lifting(self, mu, t, entries=None):
        return self._assemble_matrix(self._trilinear_elem(self.lifting(mu, t)), entries)

    def solve(self):
        """Full-order BDF time loop (host, SciPy sparse LU) producing ``solutions`` and
        ``nonlinear_snapshots`` the way ``OneDimensionalSolver.solve`` does (fom/base.py:693-831):
        snapshots hold the homogeneous part, ``fom`` adds the lifting."""
        from scipy.sparse.linalg import spsolve

        from ..storage import SolutionsStorage

        mu, dt = self.mu, self.dt
        u_n = np.zeros(self.Nh)
        u_n1 = None
        t = 0.0
        ts, homog, full, xs = [], [], [], []
        self.nonlinear_snapshots = [np.zeros(1)]
        for step in range(self.nt):
            t += dt
            bdf = 1.5 if (self.bdf2 and step > 0) else 1.0
            u_star = u_n if (u_n1 is None or not self.bdf2) else 2.0 * u_n - u_n1
            M = self.assemble_mass(mu, t)
            Nmat = self.assemble_trilinear(mu, t, u_star)
            K = bdf * M + dt * (self.assemble_stiffness(mu, t) + self.assemble_convection(mu, t) + Nmat
                                + self.assemble_nonlinear_lifting(mu, t))
            past = u_n if (u_n1 is None or not self.bdf2 or step == 0) else 2.0 * u_n - 0.5 * u_n1
            rhs = M.dot(past) + dt * self.assemble_lifting(mu, t)
            rhs[0] = rhs[-1] = 0.0
            u = spsolve(K.tocsc(), rhs)
            self.nonlinear_snapshots.append(Nmat.data.copy())
            u_n1, u_n = u_n, u
            ts.append(t)
            homog.append(u.copy())
            full.append(u + self.lifting(mu, t))
            xs.append(self.x_at(mu, t).reshape(-1, 1))
        self.solutions = SolutionsStorage(ts=ts, mu=mu, domain=np.hstack(xs), fom=np.array(full).T,
                                          snapshots=np.array(homog).T)

    def assemble_lifting(self, mu, t, entries=None):
        x = self.x_at(mu, t)
        h = self._h(mu, t)
        L = self._L(mu, t)
        gdot = -mu["delta"] * mu["omega"] * np.cos(mu["omega"] * t) * x / L
        fa, fb = gdot[:-1], gdot[1:]
        elem = np.stack([h / 6.0 * (2.0 * fa + fb), h / 6.0 * (fa + 2.0 * fb)], axis=1)
        return self._assemble_vector(elem, entries)


def _p1_closed_form(self, mus, ts):
    """Scalar tables of the closed-form P1 operators for every (t, mu): what ``rt_p1_local_assembly`` needs instead of
    per-(mu, t) assembly callbacks.  (nt, n_mu) arrays: ``h`` cell size, ``alpha`` diffusivity (stiffness factor),
    ``lift`` amplitude of the lifting ramp g = lift x / L, ``lift_dot`` amplitude of its time derivative's ramp."""
    ts = np.asarray(ts, dtype=float)
    L = np.empty((ts.size, len(mus)))
    for j, mu in enumerate(mus):
        try:
            col = np.broadcast_to(np.asarray(self._L(mu, ts), dtype=float), ts.shape)
        except Exception:  # a scale function that does not take arrays
            col = np.array([self._L(mu, t) for t in ts])
        L[:, j] = col
    a0 = np.array([mu["alpha_0"] for mu in mus])[None, :]
    out = dict(nx=self.nx, h=L / self.nx, alpha=a0 * (1.0 + ts[:, None] ** 2))
    if all(("delta" in mu and "omega" in mu) for mu in mus):
        d = np.array([mu["delta"] for mu in mus])[None, :]
        w = np.array([mu["omega"] for mu in mus])[None, :]
        out["lift"] = d * np.sin(w * ts[:, None])
        out["lift_dot"] = -d * w * np.cos(w * ts[:, None])
    return out


MockSolver.p1_closed_form = _p1_closed_form


class AffineBurgers:
    """Synthetic Burgers-type FOM whose operators are affine in fixed value vectors on ONE CSR pattern
    (pentadiagonal), the form the device sweep (rt_rom_bdf_sweep) consumes:

        A(mu)      = alpha          * A0           C(mu, t) = beta  * sin(omega t) * C0
        N(u)       = diag(u) T                     Nhat(mu, t) = delta * cos(omega t) * N0
        f_g(mu, t) = sin(omega t) f0 + delta t f1

    It serves the same ``assemble_*`` callbacks as the FEniCS solvers (rom.py:877-929 call sites), so the
    reference loop / the oracle / RomConstructorNonlinear can run on it, and ``descriptor`` lays the same
    data out for the device.  ``mu`` keys: alpha, beta, delta, omega.  Lifting function g_h = 0."""

    RUNTIME_PROCESS = False
    exact_solution = None

    def __init__(self, N, nt, dt, bdf2=True, seed=0):
        rng = np.random.RandomState(seed)
        self.Nh, self.nt, self._dt, self.bdf2 = int(N), int(nt), float(dt), bool(bdf2)
        self.BDF_SCHEME = "2" if bdf2 else "1"
        self.domain = {"nt": self.nt, "T": self.nt * self._dt, "nx": self.Nh - 1, "L0": 1.0}
        self.is_setup = True
        offs = [-2, -1, 0, 1, 2]
        rows = np.concatenate([np.arange(max(0, -o), min(N, N - o)) for o in offs])
        cols = np.concatenate([np.arange(max(0, -o), min(N, N - o)) + o for o in offs])
        pat = csr_matrix((np.ones(rows.size), (rows, cols)), shape=(N, N))
        pat.sort_indices()
        self.indptr, self.indices = pat.indptr.astype(np.int64), pat.indices.astype(np.int64)
        rr = np.repeat(np.arange(N), np.diff(self.indptr))
        off = self.indices - rr
        x = (rr + 0.5) / N
        self.mass = np.where(off == 0, 1.0, np.where(np.abs(off) == 1, 0.15, 0.02)) * (1.0 + 0.1 * np.sin(3 * x))
        self.A0 = np.where(off == 0, 2.5, np.where(np.abs(off) == 1, -1.0, -0.25)) * (1.0 + 0.3 * x) * 40.0
        self.C0 = np.sign(off) * (1.0 / np.maximum(np.abs(off), 1)) * (1.0 + 0.2 * np.cos(2 * x)) * 3.0
        self.N0 = (0.3 * rng.standard_normal(rows.size) + np.where(off == 0, 1.0, 0.0)) * 2.0
        self.T = np.sign(off) * (0.5 + 0.1 * rng.standard_normal(rows.size)) * 5.0
        xs = (np.arange(N) + 0.5) / N
        self.f = np.stack([np.sin(np.pi * xs), xs * (1 - xs) * 4.0])

    def setup(self):
        pass

    def update_parametrization(self, mu):
        self.mu = dict(mu)

    @property
    def dt(self):
        return self._dt

    def _csr(self, vals):
        return csr_matrix((vals, self.indices, self.indptr), shape=(self.Nh, self.Nh))

    # coefficient functions (shared by the host callbacks and the device descriptor)
    @staticmethod
    def thetas(mu, t):
        return np.array([mu["alpha"], mu["beta"] * np.sin(mu["omega"] * t), mu["delta"] * np.cos(mu["omega"] * t)])

    @staticmethod
    def phis(mu, t):
        return np.array([np.sin(mu["omega"] * t), mu["delta"] * t])

    def assemble_mass(self, mu, t):
        return self._csr(self.mass)

    def assemble_stiffness(self, mu, t):
        return self._csr(self.thetas(mu, t)[0] * self.A0)

    def assemble_convection(self, mu, t):
        return self._csr(self.thetas(mu, t)[1] * self.C0)

    def assemble_nonlinear_lifting(self, mu, t):
        return self._csr(self.thetas(mu, t)[2] * self.N0)

    def assemble_trilinear(self, mu, t, u_n):
        rr = np.repeat(np.arange(self.Nh), np.diff(self.indptr))
        return self._csr(np.asarray(u_n)[rr] * self.T)

    def assemble_lifting(self, mu, t):
        return self.phis(mu, t) @ self.f

    def lifting(self, mu, t):
        return np.zeros(self.Nh)

    def x_at(self, mu, t):
        return np.linspace(0.0, 1.0, self.Nh)

    def descriptor(self, mus):
        """Arrays for romtime_amd.sweep.rom_bdf_sweep: term values (3 x nnz), coefficients per step/mu."""
        ts = self._dt * np.arange(1, self.nt + 1)
        term_coef = np.array([[self.thetas(mu, t) for mu in mus] for t in ts])  # nt x n_mu x 3
        rhs_coef = np.array([[self.phis(mu, t) for mu in mus] for t in ts])     # nt x n_mu x 2
        return dict(indptr=self.indptr, indices=self.indices, mass=self.mass,
                    terms=np.stack([self.A0, self.C0, self.N0]), term_coef=term_coef, tril=self.T,
                    rhs_terms=self.f, rhs_coef=rhs_coef, dt=self._dt, bdf2=self.bdf2)
