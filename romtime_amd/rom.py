"""Reduced-order-model constructors (class surface of rom/rom.py:34-974) with the reduction
algebra on the MI355X.

  * reduced basis: two-level POD tree walk              -> ``pod.orth``
  * ``to_rom``:  A_N = V^T (A V),  f_N = V^T f            -> rt_project_csr / rt_gemm_tn
  * ``to_fom_vector``: u_h = V u_N                        -> rt_gemm_nn
  * reduced solve (the reference's GMRES(20) on a dense r x r system, rom.py:36,492)
                                                          -> rt_dense_solve_batched (pivoted LU)
  * hyper-reduced operators                               -> the (M)DEIM classes
The reduced basis V lives on the device for the whole online loop; the FOM callbacks
(FEniCS or any duck type) stay on the host, so one time step moves each assembled
operator host->device once.  All three classes accept the 5-argument
``assemble_system(mu, t, bdf, uh, uh_n1)`` / ``assemble_system_rhs(mu, t, MN, uN_n, uN_n1)``
calls that ``solve`` issues (rom.py:487-488); in the reference only the Nonlinear class does.
"""
from __future__ import annotations

from copy import deepcopy

import numpy as np
import torch

from . import ops
from .base import Reductor
from .conventions import BDF, OperatorType, PistonParameters, RomParameters, Stage, Treewalk, TreewalkNonlinear
from .storage import RomSolutionsStorage
from .utils import CsrPattern, bilinear_to_csr, function_to_array, functional_to_array, is_matrix_like

_HYPER_SLOTS = {
    OperatorType.FORCING: "deim_fh",
    OperatorType.LIFTING: "deim_fgh",
    OperatorType.RHS: "deim_rhs",
    OperatorType.MASS: "mdeim_Mh",
    OperatorType.STIFFNESS: "mdeim_Ah",
    OperatorType.CONVECTION: "mdeim_Ch",
    OperatorType.TRILINEAR: "mdeim_Nh",
    OperatorType.NONLINEAR_LIFTING: "mdeim_Nh_hat",
}


class RomConstructor(Reductor):
    # kept for callers that read it; the device solve is direct and needs no tolerances
    GMRES_OPTIONS = dict(atol=1e-10, tol=1e-10, maxiter=1e6)

    def __init__(self, fom, grid, name=None) -> None:
        super().__init__(grid=grid)
        self.fom = fom
        self.name = name
        self.basis = None
        self.basis_nonlinear = None
        self.solutions = dict()
        self.errors = dict()
        self.exact = dict()
        for slot in _HYPER_SLOTS.values():
            setattr(self, slot, None)
        self._V_dev = None
        self._patterns = []

    # ---- basis bookkeeping ---------------------------------------------------------------
    @property
    def N(self):
        return self.basis.shape[1]

    @property
    def shape(self):
        return self.basis.shape

    @property
    def timesteps(self):
        return self.solutions.ts

    def _V(self) -> torch.Tensor:
        if self._V_dev is None or self._V_dev[0] is not self.basis:
            self._V_dev = (self.basis, ops.to_device(np.ascontiguousarray(self.basis)))
        return self._V_dev[1]

    def _pattern(self, A) -> CsrPattern:
        for pat in self._patterns:
            if pat.shape == A.shape and pat.indices_host.size == A.indices.size and \
                    np.array_equal(pat.indptr_host, A.indptr) and np.array_equal(pat.indices_host, A.indices):
                return pat
        pat = CsrPattern.from_csr(A)
        self._patterns = (self._patterns + [pat])[-8:]
        return pat

    # ---- projections (rom.py:97-158) -----------------------------------------------------
    def to_fom_vector(self, uN):
        """u_h = V u_N."""
        return ops.gemm_nn(self._V(), ops.to_device(np.asarray(uN, dtype=float))).cpu().numpy()

    def to_rom_vector(self, uh):
        """u_N = V^T u_h."""
        return ops.gemm_tn(self._V(), ops.to_device(function_to_array(uh))).cpu().numpy()

    def _to_rom_dev(self, oph) -> torch.Tensor:
        V = self._V()
        if is_matrix_like(oph):
            A = bilinear_to_csr(oph)
            indptr, indices = self._pattern(A).device()
            return ops.project_csr(indptr, indices, ops.to_device(A.data), V)
        return ops.gemm_tn(V, ops.to_device(functional_to_array(oph)))

    def to_rom(self, oph):
        """FOM operator -> ROM operator: V^T A V for matrices, V^T f for vectors (rom.py:135-158)."""
        return self._to_rom_dev(oph).cpu().numpy()

    def load_from_basis(self, basis, mu_space):
        self.basis = deepcopy(basis)
        mu_space[Stage.ONLINE] = []
        mu_space[Stage.VALIDATION] = []
        self.mu_space = deepcopy(mu_space)

    def truncate(self, n):
        """ROM with the last ``n`` basis vectors removed (S-ROM -> ROM, rom.py:169-198)."""
        truncated = self.__class__(fom=self.fom, grid=self.grid, name=self.name)
        truncated.setup(rnd=self.random_state)
        N = self.N
        assert n < N, "You want to remove too many modes from S-ROM to create ROM."
        truncated.basis = self.basis[:, : N - n]
        truncated.mu_space = deepcopy(self.mu_space)
        truncated.report = deepcopy(self.report)
        truncated.report[Stage.OFFLINE][Treewalk.BASIS_FINAL] = truncated.N
        return truncated

    def setup(self, rnd):
        super().setup(rnd=rnd)
        self.algebraic_solver = self.create_algebraic_solver()

    def add_hyper_reductor(self, reductor, which):
        """Attach a COPY of an (M)DEIM object for one operator (rom.py:213-252)."""
        if which not in _HYPER_SLOTS:
            raise NotImplementedError(f"Which is this reductor? {which}")
        setattr(self, _HYPER_SLOTS[which], reductor.copy())

    def project_reductors(self):
        """Project every attached collateral basis onto V (rom.py:254-274)."""
        for slot in _HYPER_SLOTS.values():
            red = getattr(self, slot)
            if red:
                red.project_basis(V=self.basis)

    # ---- offline: reduced basis (rom.py:276-412) -----------------------------------------
    def build_reduced_basis(self, num_snapshots=None, mu_space=None, num_basis=None, tolerances=dict()):
        if num_snapshots:
            space = self.build_sampling_space(num=num_snapshots, rnd=self.random_state)
        elif mu_space:
            space = mu_space
        else:
            raise NotImplementedError("You need to provide a number of mu-snapshots or a space.")
        fom = self.fom
        if fom.is_setup == False:  # noqa: E712
            fom.setup()
        off = self.report[Stage.OFFLINE]
        tol_t = tolerances.get(RomParameters.TOL_TIME, None)
        tol_mu = tolerances.get(RomParameters.TOL_MU, None)
        fom_solutions = dict()
        from . import walks

        # The time-level PODs of all parameters (solution snapshots and, for the nonlinear problems, the snapshots of
        # the state-dependent operator) are ONE sequence of independent PODs with the same truncation rule: every set is
        # uploaded once when the FOM has produced it and goes through the device's POD lanes; the bases stay on the
        # device for the mu-level PODs (walks.py).
        tags = []

        def time_level_sets():
            for mu in space:
                mu_idx, mu = self.add_mu(mu=mu, step=Stage.OFFLINE)
                fom.setup()
                fom.update_parametrization(mu)
                fom.solve()
                fom_solutions[mu_idx] = fom.solutions.fom.copy()
                tags.append((mu_idx, False))
                yield walks.upload(fom.solutions.snapshots)            # time-level POD, normalised (rom.py:335)
                nl = getattr(fom, "nonlinear_snapshots", None)
                if nl is not None and len(nl) > 1:
                    snaps = np.array(nl[1:]).T  # first one is zero (initial condition), rom.py:345
                    snaps[0, :] = 0.0
                    tags.append((mu_idx, True))
                    yield walks.upload(snaps)
                if getattr(fom, "RUNTIME_PROCESS", False) and hasattr(fom, "save_probes"):
                    fom.save_probes(name=f"probes_offline_fom_{mu_idx}.csv")

        per_mu, per_mu_nl, width = [], [], dict()
        for i, out in enumerate(walks.pod_sequence(time_level_sets(), tol=tol_t)):
            mu_idx, nonlinear = tags[i]
            if not nonlinear:
                per_mu.append(out["Q"])
                width[mu_idx] = out["Q"].shape[1]
                off[Treewalk.SPECTRUM_TIME][mu_idx] = out["s"]
                off[Treewalk.ENERGY_TIME][mu_idx] = out["energy"]
                off[Treewalk.BASIS_TIME][mu_idx] = out["Q"].shape[1]
            else:
                per_mu_nl.append(out["Q"])
                off[TreewalkNonlinear.SPECTRUM_TIME][mu_idx] = out["s"]
                off[TreewalkNonlinear.ENERGY_TIME][mu_idx] = out["energy"]
                off[TreewalkNonlinear.BASIS_TIME][mu_idx] = width[mu_idx]  # sic: rom.py:359-361
        top = walks.pod_of_stack(per_mu, num=num_basis, tol=tol_mu, normalize=False)
        en_mu = top["energy"]
        off[Treewalk.BASIS_AFTER_WALK] = top["stacked_columns"]
        off[Treewalk.SPECTRUM_MU] = top["s"]
        off[Treewalk.ENERGY_MU] = en_mu
        off[Treewalk.BASIS_FINAL] = top["Q"].shape[1]
        self.basis = top["Q"].cpu().numpy()
        if per_mu_nl:
            top_nl = walks.pod_of_stack(per_mu_nl, normalize=False)
            off[TreewalkNonlinear.BASIS_AFTER_WALK] = top_nl["stacked_columns"]
            off[TreewalkNonlinear.SPECTRUM_MU] = top_nl["s"]
            off[TreewalkNonlinear.ENERGY_MU] = top_nl["energy"]
            off[TreewalkNonlinear.BASIS_FINAL] = top_nl["Q"].shape[1]
            self.basis_nonlinear = top_nl["Q"].cpu().numpy()
        assert self.N != 0, (
            "(ROM) There are no basis vectors. \n See tolerance according to mu-energy: "
            f"{tolerances.get(RomParameters.TOL_MU)} < {en_mu}"
        )
        return fom_solutions

    # ---- online (rom.py:414-555) ----------------------------------------------------------
    def create_algebraic_solver(self):
        """``solver(A=K_N, b=b_N) -> (u_N, info)`` like the partial(gmres) of rom.py:414-425;
        direct pivoted-LU on the device, ``info`` = 0 or 2 (singular pivot)."""

        def solver(A, b):
            dev = isinstance(A, torch.Tensor)
            x, info = ops.dense_solve(ops.to_device(A), ops.to_device(b))
            return (x if dev else x.cpu().numpy()), int(info[0].item())

        return solver

    def runtime_process(self, u=None, mu=None, t=None):
        pass

    def _lifting_on_grid(self, mu, t):
        """(x, g_h) at time t: nodal coordinates of the moved mesh and the lifting function on it
        (rom.py:507-515).  Duck-typed FOMs may provide ``lifting(mu, t)`` / ``x_at(mu, t)``."""
        fom = self.fom
        if hasattr(fom, "lifting"):
            return np.asarray(fom.x_at(mu, t)).reshape(-1, 1), np.asarray(fom.lifting(mu, t))
        fom.move_mesh(mu=mu, t=t)
        x = fom.x.copy()
        g, _, _ = fom.create_lifting_operator(mu=mu, t=t, L=fom.L)
        fom.move_mesh(back=True)
        return x, function_to_array(fom.interpolate_func(g, fom.V, mu, t))

    def _exact_on_grid(self, mu, t):
        fom = self.fom
        if callable(fom.exact_solution):
            return np.asarray(fom.exact_solution(fom.x_at(mu, t), t, **mu))
        import fenics

        ue = fenics.Expression(fom.exact_solution, degree=1, t=t, **mu)
        return function_to_array(fom.interpolate_func(ue, fom.V, mu, t)).copy()

    def solve(self, mu, step):
        """BDF1/BDF2 time loop in the reduced space; zero initial condition (rom.py:430-555)."""
        idx_mu, mu = self.add_mu(mu=mu, step=step)
        fom = self.fom
        track_error = fom.exact_solution is not None
        errors, exact = [], dict()
        dt = fom.dt
        t = 0.0
        V = self._V()
        dev = V.device
        uN_n = torch.zeros(self.N, dtype=torch.float64, device=dev)
        uh = np.zeros(self.basis.shape[0])
        bdf2 = fom.BDF_SCHEME == BDF.TWO
        uh_n1 = None
        uN_n1 = torch.zeros_like(uN_n) if bdf2 else None
        timesteps, fom_cols, rom_cols, domains = [], [], [], []
        for timestep in range(fom.domain["nt"]):
            t += dt
            timesteps.append(t)
            bdf = 1.5 if (bdf2 and timestep > 0) else 1.0
            MN, KN = self.assemble_system(mu, t, bdf, uh, uh_n1)
            bN = self.assemble_system_rhs(mu, t, MN, uN_n, uN_n1)
            uN, _info = self.algebraic_solver(A=KN, b=bN)
            uN = ops.to_device(uN)
            rom_cols.append(uN)
            if bdf2:
                uN_n1 = uN_n
                uh_n1 = uh
            uN_n = uN
            uh = ops.gemm_nn(V, uN).cpu().numpy()
            x, gh = self._lifting_on_grid(mu, t)
            domains.append(x)
            uc_h = uh + gh
            fom_cols.append(uc_h)
            if track_error:
                ue_h = self._exact_on_grid(mu, t)
                exact[t] = ue_h
                errors.append(self._compute_error(u=uc_h, ue=ue_h))
        self.solutions = RomSolutionsStorage(
            ts=timesteps, mu=mu, domain=np.hstack(domains), fom=np.vstack(fom_cols).T,
            rom=torch.stack(rom_cols, dim=1).cpu().numpy(),
        )
        if track_error:
            self.errors.update({idx_mu: np.array(errors)})
            self.exact.update({idx_mu: exact})
        return idx_mu

    # ---- reduced operators: hyper-reduced if a reductor is attached, else project the FOM's -----
    def _reduced(self, slot, fom_assemble, mu, t, **kw):
        red = getattr(self, slot)
        if red:
            return ops.to_device(red.interpolate(mu=mu, t=t, which=self.ROM, **kw))
        return self._to_rom_dev(fom_assemble(mu, t, **kw) if kw else fom_assemble(mu, t))

    def _public(self, x):
        return x.cpu().numpy()

    def assemble_mass(self, mu, t):
        return self._public(self._mass(mu, t))

    def assemble_stiffness(self, mu, t):
        return self._public(self._stiffness(mu, t))

    def assemble_forcing(self, mu, t):
        return self._public(self._reduced("deim_fh", self.fom.assemble_forcing, mu, t))

    def assemble_lifting(self, mu, t):
        return self._public(self._lifting(mu, t))

    def assemble_rhs(self, mu, t):
        return self._public(self._rhs(mu, t))

    def _mass(self, mu, t):
        return self._reduced("mdeim_Mh", self.fom.assemble_mass, mu, t)

    def _stiffness(self, mu, t):
        return self._reduced("mdeim_Ah", self.fom.assemble_stiffness, mu, t)

    def _lifting(self, mu, t):
        return self._reduced("deim_fgh", self.fom.assemble_lifting, mu, t)

    def _rhs(self, mu, t):
        """forcing + lifting together (rom.py:617-640)."""
        if self.deim_rhs:
            return ops.to_device(self.deim_rhs.interpolate(mu=mu, t=t, which=self.ROM))
        return self._to_rom_dev(self.fom.assemble_forcing(mu, t)) + self._to_rom_dev(self.fom.assemble_lifting(mu, t))

    def _system_operators(self, mu, t, uh, uh_n1):
        """The reduced operators summed into K_N besides the mass matrix."""
        return self._stiffness(mu, t)

    def _source(self, mu, t):
        return self._rhs(mu, t)

    def assemble_system(self, mu, t, bdf=None, uh=None, uh_n1=None):
        """(M_N, K_N) with K_N = bdf M_N + dt (sum of reduced operators)."""
        MN = self._mass(mu, t)
        KN = (1.0 if bdf is None else bdf) * MN + self.fom.dt * self._system_operators(mu, t, uh, uh_n1)
        return MN, KN

    def assemble_system_rhs(self, mu, t, MN_mat, uN_n, uN_n1=None):
        """b_N = M_N u^n + dt f_N (BDF1) or M_N (2 u^n - u^{n-1}/2) + dt f_N (BDF2)."""
        MN = ops.to_device(MN_mat)
        u = ops.to_device(uN_n)
        if uN_n1 is not None:
            u = 2.0 * u - 0.5 * ops.to_device(uN_n1)
        return MN @ u + self.fom.dt * self._source(mu, t)


class RomConstructorMoving(RomConstructor):
    """Adds the ALE convection operator (rom.py:688-736)."""

    def _convection(self, mu, t):
        return self._reduced("mdeim_Ch", self.fom.assemble_convection, mu, t)

    def assemble_convection(self, mu, t):
        return self._public(self._convection(mu, t))

    def _system_operators(self, mu, t, uh, uh_n1):
        return self._stiffness(mu, t) + self._convection(mu, t)


class RomConstructorNonlinear(RomConstructorMoving):
    """Burgers/piston ROM: trilinear N(u*) with u* = 2 u^n - u^{n-1} and the nonlinear lifting
    operator (rom.py:739-974)."""

    PISTON_MACH_MIN = 0.15
    PISTON_MACH_MAX = 0.4

    def __init__(self, fom, grid, name=None) -> None:
        super().__init__(fom=fom, grid=grid, name=name)
        self.probe_location = getattr(fom, "probe_location", None)
        self.probes = None

    @staticmethod
    def compute_piston_mach_number(sample):
        P = PistonParameters
        return sample[P.DELTA] * sample[P.OMEGA] / sample[P.A0]

    @staticmethod
    def compute_piston_mach_number_space(grid, num, mach_min=None, mach_max=None):
        P = PistonParameters
        lo = {k: min(grid[k].support()) for k in (P.A0, P.OMEGA, P.DELTA)}
        hi = {k: max(grid[k].support()) for k in (P.A0, P.OMEGA, P.DELTA)}
        if mach_min is None:
            mach_min = lo[P.DELTA] * lo[P.OMEGA] / hi[P.A0]
        if mach_max is None:
            mach_max = hi[P.DELTA] * hi[P.OMEGA] / lo[P.A0]
        return np.linspace(start=mach_min, stop=mach_max, num=num + 1)

    def build_sampling_space(self, num, rnd=None):
        """One sample per piston-Mach-number bin, sorted by Mach number (rom.py:751-815)."""
        edges = self.compute_piston_mach_number_space(self.grid, num, self.PISTON_MACH_MIN, self.PISTON_MACH_MAX)
        open_bins = list(zip(edges, edges[1:]))
        picked = []
        for sample in Reductor.build_sampling_space(self, num=int(2e4), rnd=rnd):
            mach = self.compute_piston_mach_number(sample)
            hit = next((b for b in open_bins if b[0] <= mach <= b[1]), None)
            if hit is not None:
                sample[PistonParameters.MACH_PISTON] = mach
                picked.append(sample)
                open_bins.remove(hit)
            if not open_bins:
                break
        return sorted(picked, key=lambda s: s[PistonParameters.MACH_PISTON])

    def _trilinear(self, mu, t, uh):
        return self._reduced("mdeim_Nh", self.fom.assemble_trilinear, mu, t, u_n=uh)

    def _nonlinear_lifting(self, mu, t):
        return self._reduced("mdeim_Nh_hat", self.fom.assemble_nonlinear_lifting, mu, t)

    def assemble_trilinear(self, mu, t, uh):
        return self._public(self._trilinear(mu, t, uh))

    def assemble_nonlinear_lifting(self, mu, t):
        return self._public(self._nonlinear_lifting(mu, t))

    def _system_operators(self, mu, t, uh, uh_n1):
        u_star = uh if uh_n1 is None else 2.0 * uh - uh_n1  # rom.py:897-901
        return (self._stiffness(mu, t) + self._convection(mu, t) + self._trilinear(mu, t, u_star)
                + self._nonlinear_lifting(mu, t))

    def _source(self, mu, t):
        return self._lifting(mu, t)  # no forcing term for Burgers (rom.py:913-915)
