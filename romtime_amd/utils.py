"""Converters and small algebra helpers at the FOM -> reduction boundary.

FOM operators may arrive as dolfin ``Matrix``/``Vector`` (FEniCS host assembly, exactly what
``src/romtime/utils.py:58-93`` converts), as ``scipy.sparse`` matrices or as ndarrays; all of
them are funnelled into CSR triplets / float64 vectors on the device.
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix, issparse

from . import ops

ZERO_TOLERANCE = 1e-15  # utils.py:163


def bilinear_to_csr(matrix) -> csr_matrix:
    """dolfin Matrix (PETSc backend) / scipy sparse / dense -> scipy CSR (utils.py:76-93)."""
    if issparse(matrix):
        return matrix.tocsr()
    if isinstance(matrix, np.ndarray):
        return csr_matrix(matrix)
    try:
        import fenics  # only present on a FEniCS host

        petsc_mat = fenics.as_backend_type(matrix).mat()
    except ImportError:
        petsc_mat = matrix.mat()
    indptr, indices, data = petsc_mat.getValuesCSR()
    return csr_matrix((data, indices, indptr), shape=petsc_mat.size)


def functional_to_array(operator) -> np.ndarray:
    """dolfin Vector / array-like -> float64 ndarray (utils.py:58-73)."""
    return np.asarray(operator, dtype=np.float64)


def function_to_array(func) -> np.ndarray:
    """dolfin Function -> coefficient array (utils.py:44-55); arrays pass through."""
    if isinstance(func, np.ndarray):
        return func
    return func.vector().vec().array


def is_matrix_like(op) -> bool:
    if issparse(op):
        return True
    if isinstance(op, np.ndarray):
        return op.ndim == 2
    return hasattr(op, "mat") or type(op).__name__ == "Matrix"


def eliminate_zeros(Ah: csr_matrix) -> csr_matrix:
    """|a| <= 1e-15 becomes a structural zero (utils.py:152-168); mutates like the reference."""
    small = np.isclose(Ah.data, 0, rtol=ZERO_TOLERANCE, atol=ZERO_TOLERANCE)
    Ah.data[small] = 0
    Ah.eliminate_zeros()
    return Ah


def vector_to_csr(entries, rows, cols) -> csr_matrix:
    """Value vector on a (rows, cols) pattern -> CSR (utils.py:136-149)."""
    return csr_matrix((np.asarray(entries), (np.asarray(rows), np.asarray(cols))))


class CsrPattern:
    """A fixed sparsity pattern resident on the device: indptr/indices (int64) plus the permutation
    that takes a value vector in (rows, cols) order to CSR order (identity when rows are sorted,
    which ``get_matrix_topology`` guarantees, mdeim.py:145-149)."""

    def __init__(self, rows, cols, shape=None):
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        n_rows = int(shape[0]) if shape is not None else int(rows.max()) + 1
        self.shape = (n_rows, int(shape[1]) if shape is not None else int(cols.max()) + 1)
        order = np.argsort(rows, kind="stable")
        self.is_sorted = bool(np.all(order == np.arange(rows.size)))
        self.order = order
        counts = np.bincount(rows, minlength=n_rows)
        self.indptr_host = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.indices_host = cols[order]
        self._dev = None

    @classmethod
    def from_csr(cls, A: csr_matrix):
        self = cls.__new__(cls)
        self.shape = A.shape
        self.is_sorted = True
        self.order = None
        self.indptr_host = A.indptr.astype(np.int64)
        self.indices_host = A.indices.astype(np.int64)
        self._dev = None
        return self

    def device(self):
        if self._dev is None:
            self._dev = (ops.to_device_index(self.indptr_host), ops.to_device_index(self.indices_host))
        return self._dev

    def values(self, data):
        data = np.asarray(data, dtype=np.float64)
        return data if self.is_sorted else data[self.order]


def project_csr(Ah, V):
    """A_N = V^T (A V) on the device (utils.py:96-113). ndarray in -> ndarray out."""
    Ah = bilinear_to_csr(Ah)
    pat = CsrPattern.from_csr(Ah)
    ip, ix = pat.device()
    Vd = ops.to_device(V)
    AN = ops.project_csr(ip, ix, ops.to_device(Ah.data), Vd)
    return AN.cpu().numpy() if isinstance(V, np.ndarray) else AN


def compute_error(u, ue) -> float:
    """Discrete L2 error ||u - ue||_2 / sqrt(N) (rom/base.py:52-73)."""
    e = np.asarray(u) - np.asarray(ue)
    return float(np.linalg.norm(e, ord=2) / np.sqrt(len(u)))


def compute_rom_difference(uN, uN_srom, V_srom) -> float:
    """S-ROM error estimator ||V_s (u_s - [u; 0])||_2 / sqrt(N_h) (utils.py:173-212)."""
    uN, uN_srom = np.asarray(uN, dtype=float), np.asarray(uN_srom, dtype=float)
    padded = np.zeros_like(uN_srom)
    padded[: len(uN)] = uN
    lifted = np.asarray(V_srom) @ (uN_srom - padded)
    return float(np.linalg.norm(lifted, ord=2) / np.sqrt(len(lifted)))


def singular_to_energy(sigmas):
    ev = np.power(sigmas, 2)
    return np.cumsum(ev) / np.sum(ev)


def singular_to_pod_error(sigmas):
    ev = np.power(sigmas, 2)
    return np.sqrt(np.sum(ev) - np.cumsum(ev))


def read_pickle(path):
    import pickle

    with open(path, mode="rb") as fp:
        return pickle.load(fp)


def dump_pickle(path, obj):
    import pickle

    with open(path, mode="wb") as fp:
        pickle.dump(obj, fp)


def dump_json(path, obj):
    """``mu_space.json`` / ``setup.json`` writer (utils.py:214-218; ujson there, the same wire format)."""
    import json

    with open(path, mode="w") as fp:
        json.dump(obj, fp)


def read_json(path):
    import json

    with open(path, mode="r") as fp:
        return json.load(fp)


def dump_csv(path, obj):
    """Tabular report writer of the drivers (utils.py:228-233)."""
    import pandas as pd

    pd.DataFrame(obj).to_csv(path)
