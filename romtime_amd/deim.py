"""Discrete Empirical Interpolation on the MI355X (class surface of deim/deim.py:25-561).

What runs where:
  * snapshot POD (two-level tree walk)          -> ``pod.orth``              (rt_gram / rt_gemm_nn)
  * greedy interpolation-index selection        -> ``ops.deim_greedy``       (rt_deim_greedy)
  * theta solve  PT_U theta = f|dofs            -> ``ops.dense_solve``       (rt_dense_solve_batched)
  * interpolant  Vf theta, projection V^T Phi   -> ``ops.gemm_nn / gemm_tn`` (FP64 MFMA GEMM)
The FOM callback ``assemble(mu, t[, entries])`` stays on the host (FEniCS or any duck type).
Attributes (``basis_fom``, ``basis_rom``, ``PT_U``, ``dofs``, ``sigmas``, ``N_V``) are NumPy
arrays / lists exactly as in the reference so drivers and pickles keep working.
"""
from __future__ import annotations

from copy import deepcopy

import numpy as np

from . import ops
from .base import Reductor
from .conventions import EmpiricalInterpolation, RomParameters, Stage
from .pod import orth
from .utils import dump_pickle, functional_to_array, read_pickle


class InterpolationSelector:
    """The boolean selection operator P (N_h x m, one 1 per column) of deim.py:517-561 without the
    N_h x m dense storage.  ``P.T @ B`` / ``np.matmul(P.T, B)`` gather rows of B (the products by
    0/1 of the reference are exact, so the gather is bit-identical); ``np.asarray(P)`` densifies."""

    __array_priority__ = 1000

    def __init__(self, dofs, size, transposed=False):
        self.dofs = np.asarray(dofs, dtype=np.int64)
        self.size = int(size)
        self._t = transposed

    @property
    def shape(self):
        return (len(self.dofs), self.size) if self._t else (self.size, len(self.dofs))

    @property
    def T(self):
        return InterpolationSelector(self.dofs, self.size, not self._t)

    def __array__(self, dtype=None, copy=None):
        P = np.zeros((self.size, len(self.dofs)), dtype=dtype or np.float64)
        P[self.dofs, np.arange(len(self.dofs))] = 1.0
        return P.T if self._t else P

    def __matmul__(self, other):
        if self._t:
            return np.asarray(other)[self.dofs]
        return np.asarray(self) @ other

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if ufunc is np.matmul and method == "__call__" and inputs[0] is self and not kwargs:
            return self.__matmul__(inputs[1])
        return getattr(ufunc, method)(*[np.asarray(x) if x is self else x for x in inputs], **kwargs)


class DiscreteEmpiricalInterpolation(Reductor):
    TYPE = EmpiricalInterpolation.DEIM

    def __init__(self, assemble, grid=None, tree_walk_params=None, name=None) -> None:
        super().__init__(grid=grid)
        self.name = name
        self.assemble = assemble
        self.tree_walk_params = tree_walk_params
        self.N_V = None
        self.PT_U = None
        self.sigmas = None
        self.dofs = None
        self.basis_fom = None
        self.basis_rom = None
        self.snapshots = None
        self.basis_pickle_name = self._basis_file_name()
        self._dev_cache = {}

    # deim.py:77-81; the reference dereferences ``name`` unconditionally and breaks on None
    def _basis_file_name(self):
        label = "_".join(str(self.name if self.name is not None else "unnamed").lower().split())
        return f"basis_fom_{self.TYPE.lower()}_{label}.pkl"

    def __str__(self) -> str:
        return f"{self.TYPE} - {self.name}"

    __repr__ = __str__

    @property
    def Nh(self):
        return self.basis_fom.shape[0]

    @property
    def N(self):
        return self.basis_fom.shape[1]

    def _new_like(self):
        return self.__class__(assemble=self.assemble, grid=self.grid, tree_walk_params=self.tree_walk_params,
                              name=self.name)

    def copy(self):
        """Deep copy of the data structures, same FOM callback (deim.py:110-131)."""
        new = self._new_like()
        for attr in ("basis_fom", "basis_rom", "PT_U", "dofs", "errors_rom"):
            value = getattr(self, attr)
            if value is not None:
                setattr(new, attr, deepcopy(value))
        return new

    # ---- device residency of the (large) bases -----------------------------------------
    def _device(self, key, array):
        hit = self._dev_cache.get(key)
        if hit is None or hit[0] is not array:
            hit = (array, ops.to_device(array))
            self._dev_cache[key] = hit
        return hit[1]

    # ---- offline ------------------------------------------------------------------------
    def _finish_offline(self):
        dofs, P = self.build_interpolation_mesh()
        self.store_dofs(dofs)
        self.PT_U = np.matmul(P.T, self.basis_fom)  # == basis_fom[dofs, :]  (deim.py:159,212)

    def load_fom_basis(self, keep=None, basis=None):
        """Adopt a pre-computed collateral basis and rebuild dofs / PT_U (deim.py:133-164)."""
        if basis is None:
            basis = read_pickle(self.basis_pickle_name)
        if keep:
            basis = basis[:, :keep]
        self.basis_fom = basis
        self._finish_offline()

    def dump_fom_basis(self, path=None):
        if self.basis_fom is not None:
            dump_pickle(self.basis_pickle_name, obj=self.basis_fom)

    def run(self, normalize=True, mu_space=None):
        """Offline phase: tree walk, greedy, PT_U (deim.py:175-215)."""
        p = self.tree_walk_params
        Vfh, sigmas = self.tree_walk(
            ts=p[RomParameters.TS],
            num_snapshots=p[RomParameters.NUM_SNAPSHOTS],
            num_mu=p.get(RomParameters.NUM_MU, None),
            num_t=p.get(RomParameters.NUM_TIME, None),
            tol_mu=p.get(RomParameters.TOL_MU, None),
            tol_t=p.get(RomParameters.TOL_TIME, None),
            normalize=normalize,
            mu_space=mu_space,
        )
        self.basis_fom = Vfh
        self.sigmas = sigmas
        self._finish_offline()

    def store_dofs(self, dofs):
        self.dofs = [(int(dof),) for dof in dofs]

    def build_interpolation_mesh(self):
        """Greedy DEIM on the device (deim.py:517-561): returns ``(dofs, P)``."""
        Vf = self.basis_fom
        idx, _, margin = ops.deim_greedy(self._device("basis_fom", Vf))
        dofs = [int(i) for i in idx.cpu().numpy()]
        self.greedy_margin = margin.cpu().numpy()
        return dofs, InterpolationSelector(dofs, Vf.shape[0])

    def tree_walk(self, ts, normalize=True, num_mu=None, num_t=None, tol_mu=None, tol_t=None, num_snapshots=None,
                  mu_space=None):
        """POD in time for every sampled mu, then POD of the concatenation (deim.py:279-355).

        The time-level PODs are a sequence of independent small PODs: every parameter's snapshot set is assembled by
        the FOM callback on the host, uploaded once and run through the device's POD lanes (``walks.pod_sequence``);
        the per-parameter bases stay on the device, where they are concatenated for the mu-level POD.  Only the final
        basis and the spectra of the report come back.  A subclass that overrides ``walk_time`` keeps the plain loop."""
        from . import walks

        space = mu_space if mu_space else self.build_sampling_space(num=num_snapshots, rnd=self.random_state)
        off = self.report[Stage.OFFLINE]
        if type(self).walk_time is not DiscreteEmpiricalInterpolation.walk_time:
            return self._tree_walk_host(space, ts, normalize, num_mu, num_t, tol_mu, tol_t)
        indices = []

        def time_level_sets():
            for mu in space:
                mu_idx, mu = self.add_mu(step=Stage.OFFLINE, mu=mu)
                indices.append(mu_idx)
                yield walks.upload_columns([self.assemble_snapshot(mu, t) for t in ts],
                                           zero_first_row=self.TYPE == EmpiricalInterpolation.MDEIM)   # deim.py:384-389

        per_mu = []
        for i, out in enumerate(walks.pod_sequence(time_level_sets(), num=num_t, tol=tol_t, normalize=False)):
            mu_idx = indices[i]
            off.setdefault(self.SPECTRUM_TIME, {})[mu_idx] = out["s"]
            off.setdefault(self.ENERGY_TIME, {})[mu_idx] = out["energy"]
            off.setdefault(self.BASIS_TIME, {})[mu_idx] = out["Q"].shape[1]
            per_mu.append(out["Q"])
        top = walks.pod_of_stack(per_mu, num=num_mu, tol=tol_mu, normalize=normalize)
        off[self.BASIS_AFTER_WALK] = top["stacked_columns"]
        off[self.SPECTRUM_MU] = top["s"]
        off[self.ENERGY_MU] = top["energy"]
        off[self.BASIS_FINAL] = top["Q"].shape[1]
        return top["Q"].cpu().numpy(), top["s"]

    def _tree_walk_host(self, space, ts, normalize, num_mu, num_t, tol_mu, tol_t):
        """The reference's loop as it stands (deim.py:330-349): one ``walk_time`` call per parameter."""
        off = self.report[Stage.OFFLINE]
        per_mu = []
        for mu in space:
            mu_idx, mu = self.add_mu(step=Stage.OFFLINE, mu=mu)
            basis_t, sigmas_t, energy_t = self.walk_time(mu=mu, ts=ts, num=num_t, tol=tol_t, normalize=normalize)
            off.setdefault(self.SPECTRUM_TIME, {})[mu_idx] = sigmas_t
            off.setdefault(self.ENERGY_TIME, {})[mu_idx] = energy_t
            off.setdefault(self.BASIS_TIME, {})[mu_idx] = basis_t.shape[1]
            per_mu.append(basis_t)
        stacked = np.hstack(per_mu)
        off[self.BASIS_AFTER_WALK] = stacked.shape[1]
        basis, sigmas_mu, energy_mu = orth(snapshots=stacked, num=num_mu, tol=tol_mu, normalize=normalize)
        off[self.SPECTRUM_MU] = sigmas_mu
        off[self.ENERGY_MU] = energy_mu
        off[self.BASIS_FINAL] = basis.shape[1]
        return basis, sigmas_mu

    def _time_level_snapshots(self, mu, ts):
        """N_h x n_t snapshot matrix of one parameter (host; the FOM callback assembles it), deim.py:384-389."""
        snapshots = np.array([self.assemble_snapshot(mu, t) for t in ts]).T
        if self.TYPE == EmpiricalInterpolation.MDEIM:
            snapshots[0, :] = 0.0  # boundary entry does not matter (deim.py:388-389)
        return snapshots

    def walk_time(self, mu, ts, normalize=True, num=None, tol=None):
        """Time-level POD at frozen mu; never normalised at this level (deim.py:357-397)."""
        return orth(snapshots=self._time_level_snapshots(mu, ts), num=num, tol=tol, normalize=False)

    def _assemble_functional(self, mu, t):
        return functional_to_array(self.assemble(mu=mu, t=t))

    def assemble_snapshot(self, mu, t):
        return self._assemble_functional(mu, t)

    # ---- online --------------------------------------------------------------------------
    def compute_thetas(self, rhs):
        """theta = solve(PT_U, rhs) (deim.py:477-493) by pivoted LU on the device."""
        theta, info = ops.dense_solve(self._device("PT_U", self.PT_U), ops.to_device(np.asarray(rhs, dtype=float)))
        return theta.cpu().numpy()

    def _local_values(self, mu, t, **kw):
        return np.asarray(self.assemble(mu=mu, t=t, entries=self.dofs, **kw), dtype=np.float64)

    def _expand(self, thetas, which):
        if (which is None) or (which == self.FOM):
            Vf, key = self.basis_fom, "basis_fom"
        elif which == self.ROM:
            Vf, key = self.basis_rom, "basis_rom"
        else:
            raise ValueError(f"unknown basis selector {which!r}")
        return ops.gemm_nn(self._device(key, Vf), ops.to_device(thetas)).cpu().numpy()

    def _interpolate(self, mu, t, which=None):
        """Interpolant of the operator at (mu, t) in FOM or ROM coordinates (deim.py:416-452)."""
        thetas = self.compute_thetas(rhs=self._local_values(mu, t))
        approximation = self._expand(thetas, which)
        if (which == self.FOM) and (self.TYPE == EmpiricalInterpolation.MDEIM):
            approximation[0] = 1.0  # Dirichlet entry (deim.py:449-450)
        return approximation

    def interpolate(self, mu, t, which=None):
        return self._interpolate(mu=mu, t=t, which=which)

    def project_basis(self, V):
        """basis_rom = V^T basis_fom (r x m) (deim.py:495-515)."""
        self.basis_rom = ops.gemm_tn(ops.to_device(V), self._device("basis_fom", self.basis_fom)).cpu().numpy()

    def evaluate(self, ts, num=None, mu_space=None):
        """Accuracy harness: interpolation error on fresh parameters (deim.py:226-261)."""
        if mu_space:
            space = mu_space
        else:
            assert num, "Provide number of samples to test"
            space = self.build_sampling_space(num=num)
        for mu in space:
            mu_idx, mu = self.add_mu(step=Stage.ONLINE, mu=mu)
            for t in ts:
                exact = self.assemble_snapshot(mu, t)
                approx = self._interpolate(mu, t, which=self.FOM)
                self.errors_rom[mu_idx].append(self._compute_error(u=approx, ue=exact))
            self.errors_rom[mu_idx] = np.array(self.errors_rom[mu_idx])
