"""N-MDEIM: MDEIM of the state-dependent operator N(u) (class surface of deim/nonlinear.py:26-555).

Differences from MDEIM: the FOM callback takes ``u_n``; the tree walk has a third level (POD over
the reduced-basis functions psi at every time, then over time, then over mu -- all three
normalised, nonlinear.py:392-397,451-466); ``truncate`` drops trailing modes and re-runs the
greedy; the FOM-form interpolant always pins entry 0 to 1 (nonlinear.py:280-281)."""
from __future__ import annotations

from copy import deepcopy

import numpy as np

from .base import Reductor
from .conventions import EmpiricalInterpolation, RomParameters, Stage, Treewalk
from .mdeim import MatrixDiscreteEmpiricalInterpolation, sorted_topology
from .utils import bilinear_to_csr, eliminate_zeros


class MatrixDiscreteEmpiricalInterpolationNonlinear(MatrixDiscreteEmpiricalInterpolation):
    TYPE = EmpiricalInterpolation.NONLINEAR

    def __init__(self, assemble, name=None, grid=None, tree_walk_params=None):
        super().__init__(assemble, name=name, grid=grid, tree_walk_params=tree_walk_params)
        self.u_n = None

    def truncate(self, n):
        """New reductor without the last ``n`` collateral modes; dofs and PT_U are recomputed
        (nonlinear.py:49-104)."""
        N = self.N
        assert n < N, "You want to remove too many modes from S-NonlinearMDEIM to create NonlinearMDEIM."
        truncated = self.__class__(assemble=self.assemble, grid=self.grid, tree_walk_params=self.tree_walk_params,
                                   name="S-" + str(self.name))
        Reductor.setup(self=truncated, rnd=self.random_state)
        truncated.rows, truncated.cols = self.rows, self.cols
        truncated.basis_fom = self.basis_fom[:, : N - n]
        truncated._finish_offline()
        truncated.mu_space = deepcopy(self.mu_space)
        truncated.report = deepcopy(self.report)
        truncated.report[Stage.OFFLINE][Treewalk.BASIS_FINAL] = truncated.N
        return truncated

    def get_matrix_topology(self, mu, t, u_n):
        return sorted_topology(eliminate_zeros(self._assemble_matrix(mu=mu, t=t, u_n=u_n)))

    def setup(self, rnd, V=None, u_n=None):
        """Topology from one sample operator assembled with a non-constant state
        (nonlinear.py:133-157: u_n = x interpolated on the FE space ``V``).  Off a FEniCS host pass
        the state vector ``u_n`` directly."""
        Reductor.setup(self=self, rnd=rnd)
        mu = list(self.build_sampling_space(num=1))[0]
        if u_n is None:
            import fenics

            u_n = fenics.interpolate(fenics.Expression("x[0]", degree=1), V)
        self.rows, self.cols = self.get_matrix_topology(mu=mu, t=1.0, u_n=u_n)

    def run(self, u_n, mu_space=None):
        """Offline phase over the state basis ``u_n`` (N_h x N_psi) (nonlinear.py:159-212)."""
        u_n = np.asarray(u_n)
        self.u_n = u_n.reshape(u_n.shape[0], 1) if u_n.ndim == 1 else u_n
        p = self.tree_walk_params
        Vfh, sigmas = self.tree_walk(
            ts=p[RomParameters.TS],
            normalize=True,
            num_mu=p.get(RomParameters.NUM_MU, None),
            num_t=p.get(RomParameters.NUM_TIME, None),
            num_basis=p.get(RomParameters.NUM_BASIS, None),
            tol_mu=p.get(RomParameters.TOL_MU, None),
            tol_t=p.get(RomParameters.TOL_TIME, None),
            tol_basis=p.get(RomParameters.TOL_BASIS, None),
            num_snapshots=p[RomParameters.NUM_SNAPSHOTS],
            mu_space=mu_space,
        )
        self.basis_fom = Vfh
        self.sigmas = sigmas
        self._finish_offline()

    def tree_walk(self, ts, normalize=True, num_mu=None, num_t=None, num_basis=None, tol_mu=None, tol_t=None,
                  tol_basis=None, num_snapshots=None, mu_space=None):
        """Three levels (nonlinear.py:320-403), all on the device: per parameter the basis-level PODs of every time step
        run as one sequence through the POD lanes, their bases are concatenated on the device for the time-level POD,
        and the per-parameter results again for the mu level (``walks``)."""
        from . import walks

        space = mu_space if mu_space else self.build_sampling_space(num=num_snapshots, rnd=self.random_state)
        off = self.report[Stage.OFFLINE]
        per_mu = []
        for mu in space:
            mu_idx, mu = self.add_mu(step=Stage.OFFLINE, mu=mu)
            out = self._walk_time_device(mu=mu, ts=ts, num_t=num_t, tol_t=tol_t, normalize=normalize)
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

    def _walk_time_device(self, mu, ts, normalize=True, num_t=None, tol_t=None):
        """POD over psi at every t (a sequence of independent PODs: the lanes), then POD over time of the concatenated
        bases, built on the device (nonlinear.py:405-468; the reference passes num_t / tol_t to both levels)."""
        from . import walks

        u_n = self.u_n

        def basis_level_sets():
            for t in ts:
                yield walks.upload_columns([self.assemble_snapshot(mu=mu, t=t, u_n=u_n[:, i]) for i in range(u_n.shape[1])],
                                           zero_first_row=True)                     # nonlinear.py:437-443

        per_t = [out["Q"] for out in walks.pod_sequence(basis_level_sets(), num=num_t, tol=tol_t, normalize=normalize)]
        return walks.pod_of_stack(per_t, num=num_t, tol=tol_t, normalize=normalize)

    def walk_time(self, mu, ts, normalize=True, num_t=None, tol_t=None, num_basis=None, tol_basis=None):
        """POD over psi at every t, then POD over time (nonlinear.py:405-468)."""
        out = self._walk_time_device(mu=mu, ts=ts, normalize=normalize, num_t=num_t, tol_t=tol_t)
        return out["Q"].cpu().numpy(), out["s"], out["energy"]

    def assemble_snapshot(self, mu, t, u_n):
        return eliminate_zeros(self._assemble_matrix(mu, t, u_n)).data

    def _assemble_matrix(self, mu, t, u_n):
        return bilinear_to_csr(self.assemble(mu=mu, t=t, u_n=u_n))

    def _interpolate(self, mu, t, u_n, which=None):
        thetas = self.compute_thetas(rhs=self._local_values(mu, t, u_n=u_n))
        approximation = self._expand(thetas, which)
        if which == self.FOM:
            approximation[0] = 1.0  # nonlinear.py:280-281
        return approximation

    def interpolate(self, mu, t, u_n, which=None):
        return self._to_public(self._interpolate(mu, t, u_n, which=which), which)

    def evaluate(self, ts, funcs=None, num=None, mu_space=None):
        """Mean interpolation error over the state basis (nonlinear.py:470-540)."""
        if mu_space:
            space = mu_space
        else:
            assert num, "Provide number of samples to test"
            space = self.build_sampling_space(num=num)
        u_n = self.u_n if funcs is None else funcs
        for mu in space:
            mu_idx, mu = self.add_mu(step=Stage.ONLINE, mu=mu)
            for t in ts:
                err = 0.0
                for i in range(u_n.shape[1]):
                    psi = u_n[:, i]
                    exact = self.assemble_snapshot(mu=mu, t=t, u_n=psi)
                    approx = self._interpolate(mu=mu, t=t, u_n=psi, which=self.FOM)
                    err += self._compute_error(u=approx, ue=exact)
                self.errors_rom[mu_idx].append(err / u_n.shape[1])
            self.errors_rom[mu_idx] = np.array(self.errors_rom[mu_idx])
