"""CPU stand-ins for ``romtime_amd.ops`` so that the HOST LOGIC of the drop-in classes (truncation
rules, tree-walk bookkeeping, BDF loop, pattern handling) can be exercised without a GPU.

Test infrastructure only: installed by the ``cpu_ops`` fixture via monkeypatch; the product never
routes here (without a GPU the real ops raise).  The stand-ins are the oracle's arithmetic."""
import numpy as np
import torch
from scipy.sparse import csr_matrix

from oracle import romtime_oracle as oracle


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def to_device(a, device=None):
    if isinstance(a, torch.Tensor):
        return a.to(torch.float64)
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2 and a.flags.f_contiguous and not a.flags.c_contiguous:
        return torch.from_numpy(np.ascontiguousarray(a.T)).T
    return torch.from_numpy(np.ascontiguousarray(a))


def to_device_index(a, device=None):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.int64)))


def gram(X, out=None):
    G = (X.T @ X).contiguous()
    if out is not None:
        out.copy_(G)
        return out
    return G


def gram_scale(G, normalize):
    cn = torch.sqrt(torch.diagonal(G)).clone()
    flag = torch.tensor([0 if bool((cn > 0).all()) else 1], dtype=torch.int32)
    if normalize:
        G /= cn[:, None] * cn[None, :]
    return cn, flag


def gemm_tn(A, B):
    return A.T @ B


def gemm_nn(X, T, out=None, alpha=1.0, beta=0.0):
    if out is None:
        return alpha * (X @ T) if alpha != 1.0 else X @ T
    out.copy_(alpha * (X @ T) if beta == 0.0 else beta * out + alpha * (X @ T))  # beta = 0: `out` may be uninitialised
    return out


def rank_update(Ysrc, X, T, alpha=-1.0, colscale=None, out=None):
    base = Ysrc * colscale[None, :] if colscale is not None else Ysrc
    res = base + alpha * (X @ T)
    if out is not None:
        out.copy_(res)
        return out
    return res


def deim_greedy(Phi, want_margin=True):
    dofs, PT_U, margin = oracle.deim_greedy(_np(Phi))
    return torch.from_numpy(dofs), torch.from_numpy(np.ascontiguousarray(PT_U)), torch.from_numpy(margin)


def _csr(indptr, indices, data):
    return csr_matrix((_np(data), _np(indices), _np(indptr)))


def csr_spmm(indptr, indices, data, V):
    return torch.from_numpy(_csr(indptr, indices, data).dot(_np(V)))


def project_csr(indptr, indices, data, V):
    N = indptr.numel() - 1
    A = csr_matrix((_np(data), _np(indices), _np(indptr)), shape=(N, N))
    return torch.from_numpy(oracle.project_csr(A, _np(V)))


def project_csr_batched(indptr, indices, data_batch, V):
    N = indptr.numel() - 1
    D = _np(data_batch)
    out = [oracle.project_csr(csr_matrix((D[:, b], _np(indices), _np(indptr)), shape=(N, N)), _np(V))
           for b in range(D.shape[1])]
    return torch.from_numpy(np.array(out))


def dense_solve(K, b):
    Kn, bn = _np(K), _np(b)
    if Kn.ndim == 2:
        return torch.from_numpy(np.linalg.solve(Kn, bn)), torch.zeros(1, dtype=torch.int32)
    x = np.linalg.solve(Kn, bn[..., None])[..., 0]
    return torch.from_numpy(x), torch.zeros(Kn.shape[0], dtype=torch.int32)


def backproject_weights(Z, lam, colnorm=None):
    k = Z.shape[1]
    sig = lam[:k].clamp_min(0.0).sqrt()
    inv = torch.where(sig > 0, 1.0 / sig, torch.zeros_like(sig))
    return ((Z / colnorm[:, None] if colnorm is not None else Z) * inv[None, :]).contiguous()


def install(monkeypatch):
    from romtime_amd import ops

    for name in ("to_device", "to_device_index", "gram", "gram_scale", "gemm_tn", "gemm_nn", "rank_update", "deim_greedy",
                 "csr_spmm", "project_csr", "project_csr_batched", "dense_solve", "backproject_weights"):
        monkeypatch.setattr(ops, name, globals()[name])
