"""Matrix DEIM: the DEIM machinery applied to the value vector of a sparse operator on a fixed
(rows, cols) topology (class surface of deim/mdeim.py:17-261).

``project_basis`` is the heavy offline step: for every collateral mode i the reference rebuilds
a CSR matrix from the value vector (COO sort) and computes V^T (A_i V) (mdeim.py:171-187).  Here
the topology is uploaded once as CSR and all modes are projected by one batched device call
(rt_project_csr_batched)."""
from __future__ import annotations

from copy import deepcopy

import numpy as np
from scipy.sparse import find as get_nonzero_entries

from . import ops
from .conventions import EmpiricalInterpolation
from .deim import DiscreteEmpiricalInterpolation
from .utils import CsrPattern, bilinear_to_csr, eliminate_zeros, vector_to_csr


def sorted_topology(Ah):
    """(rows, cols) of the nonzeros, stable-sorted by row, so that ``data[i]`` of the CSR value
    vector sits at ``(rows[i], cols[i])`` (mdeim.py:141-149)."""
    rows, cols, _ = get_nonzero_entries(Ah)
    pairs = sorted(zip(rows, cols), key=lambda rc: rc[0])
    return [int(rc[0]) for rc in pairs], [int(rc[1]) for rc in pairs]


class MatrixDiscreteEmpiricalInterpolation(DiscreteEmpiricalInterpolation):
    TYPE = EmpiricalInterpolation.MDEIM

    def __init__(self, assemble, name=None, grid=None, tree_walk_params=None):
        super().__init__(assemble=assemble, name=name, grid=grid, tree_walk_params=tree_walk_params)
        self.rows = None
        self.cols = None
        self._pattern = None

    def copy(self):
        new = super().copy()
        for attr in ("rows", "cols"):
            if getattr(self, attr) is not None:
                setattr(new, attr, deepcopy(getattr(self, attr)))
        new.N_V = self.N_V
        return new

    def setup(self, rnd):
        """Reset the report and read the matrix topology off one sample operator (mdeim.py:78-100)."""
        super().setup(rnd=rnd)
        mu = list(self.build_sampling_space(num=1))[0]
        self.rows, self.cols = self.get_matrix_topology(mu=mu, t=1.0)

    def get_entry(self, idx):
        return self.rows[idx], self.cols[idx]

    def store_dofs(self, dofs):
        self.dofs = [self.get_entry(int(dof)) for dof in dofs]

    def get_matrix_topology(self, mu, t):
        return sorted_topology(eliminate_zeros(self._assemble_matrix(mu, t)))

    def pattern(self) -> CsrPattern:
        if self._pattern is None or self._pattern_src is not self.rows:
            self._pattern = CsrPattern(self.rows, self.cols)
            self._pattern_src = self.rows
        return self._pattern

    def project_basis(self, V):
        """basis_rom[:, i] = flatten(V^T A_i V), A_i the i-th collateral mode on the topology
        (mdeim.py:153-192); N_V = r is stored for the online reshape."""
        self.N_V = V.shape[1]
        pat = self.pattern()
        if pat.shape[0] != V.shape[0]:
            pat = CsrPattern(self.rows, self.cols, shape=(V.shape[0], V.shape[0]))
        indptr, indices = pat.device()
        modes = self._device("basis_fom", self.basis_fom)
        if not pat.is_sorted:
            modes = modes[ops.to_device_index(pat.order)]
        AN = ops.project_csr_batched(indptr, indices, modes, ops.to_device(V))  # (m, r, r)
        self.basis_rom = AN.reshape(AN.shape[0], -1).T.contiguous().cpu().numpy()

    def assemble_snapshot(self, mu, t):
        return eliminate_zeros(self._assemble_matrix(mu, t)).data

    def _assemble_matrix(self, mu, t):
        return bilinear_to_csr(self.assemble(mu=mu, t=t))

    def _to_public(self, approximation, which):
        if which == self.ROM:
            return approximation.reshape((self.N_V, self.N_V))
        return vector_to_csr(entries=approximation, rows=self.rows, cols=self.cols)

    def interpolate(self, mu, t, which=None):
        """(N_V, N_V) ndarray for ``which=ROM``, otherwise a scipy CSR matrix (mdeim.py:230-261)."""
        return self._to_public(super()._interpolate(mu, t, which=which), which)
