"""Device operators of the POD / (M)DEIM / reduced-solve path (thin wrappers over the C ABI).

Inputs are float64 CUDA ``torch.Tensor``s (torch is the allocator / stream provider only);
``to_device`` moves NumPy data over PCIe preserving its C/F memory order so that no host
transpose is ever made.  Every function raises if the HIP library or the GPU is absent.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import COL_MAJOR, ROW_MAJOR, Context, RomtimeHipError

_p = C.c_void_p


def _ptr(t):
    return _p(t.data_ptr()) if t is not None else _p(None)


def _need_gpu():
    if not torch.cuda.is_available():
        raise RomtimeHipError("no MI355X visible: romtime_amd's hot path runs on the GPU only (there is no CPU fallback)")


def to_device(a, device=None) -> torch.Tensor:
    """NumPy (or tensor) -> float64 CUDA tensor with the same logical shape and memory order."""
    _need_gpu()
    if isinstance(a, torch.Tensor):
        t = a.to(dtype=torch.float64)
        return t.cuda(device) if not t.is_cuda else t
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2 and a.flags.f_contiguous and not a.flags.c_contiguous:
        return torch.from_numpy(np.ascontiguousarray(a.T)).cuda(device).T
    return torch.from_numpy(np.ascontiguousarray(a)).cuda(device)


def to_device_index(a, device=None) -> torch.Tensor:
    _need_gpu()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.int64))).cuda(device)


def _layout(t: torch.Tensor):
    """(tensor, ld, layout) for a 2-D float64 CUDA tensor; copies only if it has no unit stride."""
    if t.dtype != torch.float64 or not t.is_cuda or t.dim() != 2:
        raise RomtimeHipError("expected a 2-D float64 CUDA tensor")
    r, c = t.shape
    s0, s1 = t.stride()
    if s1 == 1 and (s0 >= c or r == 1):
        return t, max(s0, c), ROW_MAJOR
    if s0 == 1 and (s1 >= r or c == 1):
        return t, max(s1, r), COL_MAJOR
    t = t.contiguous()
    return t, c, ROW_MAJOR


def gram(X: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """G = X^T X (n x n). rt_gram.  ``out``: optional contiguous n x n tensor to write into."""
    ctx = Context.current()
    X, ld, lay = _layout(X)
    N, n = X.shape
    G = out if out is not None else torch.empty((n, n), dtype=torch.float64, device=X.device)
    assert G.shape == (n, n) and G.is_contiguous()
    ctx.check(ctx.lib.rt_gram(ctx.handle, _ptr(X), N, n, ld, lay, _ptr(G)), "rt_gram")
    return G


def gram_scale(G: torch.Tensor, normalize: bool):
    """In place: returns (colnorm, status_flag tensor). rt_gram_scale."""
    ctx = Context.current()
    n = G.shape[0]
    assert G.is_contiguous() and G.shape == (n, n)
    colnorm = torch.empty(n, dtype=torch.float64, device=G.device)
    flag = torch.zeros(1, dtype=torch.int32, device=G.device)
    ctx.check(ctx.lib.rt_gram_scale(ctx.handle, _ptr(G), n, _ptr(colnorm), int(bool(normalize)), _ptr(flag)),
              "rt_gram_scale")
    return colnorm, flag


def pod_orth(X: torch.Tensor, num=None, tol=None, normalize=True, q_cols=None):
    """The composite C entry point rt_pod_orth (the whole of pod.py:7-62 in one call): returns
    ``(Q[:, :r] device, s host, energy host, levels)``.  The Python drop-in ``orth`` uses ``pod.pod_device`` instead (it
    overlaps the eigenvalue fetch with the back-projection and handles process groups); this wrapper exists for parity
    tests of what a non-Python host gets."""
    ctx = Context.current()
    X, ld, lay = _layout(X)
    N, n = X.shape
    cap = int(q_cols) if q_cols is not None else (int(min(num, n)) if (num and not tol) else min(N, n))
    Q = torch.empty((N, max(cap, 1)), dtype=torch.float64, device=X.device)
    length = min(N, n)
    s, energy = np.empty(length), np.empty(length)
    r, levels = C.c_int64(0), C.c_int(0)
    rc = ctx.lib.rt_pod_orth(ctx.handle, _ptr(X), N, n, ld, lay, int(num or 0), float(tol or 0.0), int(bool(normalize)),
                             _ptr(Q), cap, C.byref(r), s.ctypes.data, energy.ctypes.data, C.byref(levels))
    if rc == _lib.WARN_ZERO_NORM:
        raise ValueError("array must not contain infs or NaNs (zero-norm snapshot with normalize=True)")
    ctx.check(rc, "rt_pod_orth")
    return Q[:, : r.value], s, energy, levels.value


def backproject_weights(Z: torch.Tensor, lam: torch.Tensor, colnorm: torch.Tensor = None) -> torch.Tensor:
    """D^-1 Z S^-1 (n x k) from the device eigenvalues ``lam`` (first k used).  rt_pod_backproject_weights."""
    ctx = Context.current()
    Z = Z.contiguous()
    n, k = Z.shape
    out = torch.empty_like(Z)
    ctx.check(ctx.lib.rt_pod_backproject_weights(ctx.handle, _ptr(Z), n, k, _ptr(colnorm), _ptr(lam), _ptr(out)),
              "rt_pod_backproject_weights")
    return out


def gemm_tn(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """C = A^T B with the contraction over the rows (DoFs). rt_gemm_tn."""
    ctx = Context.current()
    if B.dim() == 1:
        return gemm_tn(A, B.unsqueeze(1)).squeeze(1)
    same = A is B
    A, lda, la = _layout(A)
    if same:
        B, ldb, lb = A, lda, la
    else:
        B, ldb, lb = _layout(B)
    N, m = A.shape
    N2, n = B.shape
    if N != N2:
        raise RomtimeHipError(f"gemm_tn: row counts differ ({N} vs {N2})")
    Cm = torch.empty((m, n), dtype=torch.float64, device=A.device)
    ctx.check(ctx.lib.rt_gemm_tn(ctx.handle, _ptr(A), lda, la, _ptr(B), ldb, lb, N, m, n, _ptr(Cm), n), "rt_gemm_tn")
    return Cm


def gemm_nn(X: torch.Tensor, T: torch.Tensor, out: torch.Tensor = None, alpha: float = 1.0, beta: float = 0.0) -> torch.Tensor:
    """Y = X T (N x k, row-major), or ``out = beta out + alpha X T`` in place (``out``: N x k with unit column
    stride).  rt_gemm_nn / rt_gemm_nn_axpby."""
    ctx = Context.current()
    if T.dim() == 1:
        assert out is None
        return gemm_nn(X, T.unsqueeze(1)).squeeze(1)
    X, ldx, lx = _layout(X)
    T = T.contiguous()
    N, n = X.shape
    n2, k = T.shape
    if n != n2:
        raise RomtimeHipError(f"gemm_nn: inner dimensions differ ({n} vs {n2})")
    if out is None:
        if beta != 0.0:
            raise RomtimeHipError("gemm_nn: beta != 0 needs an output to accumulate into")
        Y, ldy, ly = torch.empty((N, k), dtype=torch.float64, device=X.device), k, ROW_MAJOR
    else:
        Y, ldy, ly = _layout(out)
        if Y is not out or tuple(out.shape) != (N, k):
            raise RomtimeHipError("gemm_nn: `out` must be an N x k float64 CUDA tensor with a unit stride")
    ctx.check(ctx.lib.rt_gemm_nn_axpby(ctx.handle, _ptr(X), ldx, lx, _ptr(T), k, N, n, k, float(alpha), float(beta),
                                       _ptr(Y), ldy, ly), "rt_gemm_nn_axpby")
    return Y


def rank_update(Ysrc: torch.Tensor, X: torch.Tensor, T: torch.Tensor, alpha: float = -1.0, colscale: torch.Tensor = None,
                out: torch.Tensor = None) -> torch.Tensor:
    """``out = Ysrc diag(colscale) + alpha X T`` for a tall row-major Ysrc (N x n), thin X (N x k) and T (k x n);
    ``out`` may be Ysrc itself.  k <= 64 runs the streaming kernel (rt_rank_update), larger k the GEMM epilogue."""
    ctx = Context.current()
    N, n = Ysrc.shape
    k = X.shape[1]
    if tuple(T.shape) != (k, n) or X.shape[0] != N:
        raise RomtimeHipError("rank_update: shapes do not match")
    row_major = all(t.is_contiguous() or (t.stride(1) == 1 and t.stride(0) >= t.shape[1]) for t in (Ysrc, X))
    if k > 64 or not row_major or (out is not None and not out.is_contiguous()):
        base = Ysrc * colscale[None, :] if colscale is not None else Ysrc
        out = base.clone() if (out is None and base is Ysrc) else (base if out is None else out.copy_(base))
        return gemm_nn(X, T, out=out, alpha=alpha, beta=1.0)
    T = T.contiguous()
    if out is None:
        out = torch.empty((N, n), dtype=torch.float64, device=Ysrc.device)
    ctx.check(ctx.lib.rt_rank_update(ctx.handle, _ptr(Ysrc), Ysrc.stride(0), _ptr(colscale) if colscale is not None else None,
                                     _ptr(X), X.stride(0), _ptr(T), n, N, k, n, float(alpha), _ptr(out), out.stride(0)),
              "rt_rank_update")
    return out


def deim_greedy(Phi: torch.Tensor, want_margin: bool = True):
    """(idx int64[m], PT_U [m,m], margin [m] or None), all on the device. rt_deim_greedy."""
    ctx = Context.current()
    Phi, ld, lay = _layout(Phi)
    N, m = Phi.shape
    idx = torch.empty(m, dtype=torch.int64, device=Phi.device)
    PT_U = torch.empty((m, m), dtype=torch.float64, device=Phi.device)
    margin = torch.empty(m, dtype=torch.float64, device=Phi.device) if want_margin else None
    ctx.check(ctx.lib.rt_deim_greedy(ctx.handle, _ptr(Phi), N, m, ld, lay, _ptr(idx), _ptr(PT_U), _ptr(margin)),
              "rt_deim_greedy")
    return idx, PT_U, margin


def csr_spmm(indptr, indices, data, V: torch.Tensor) -> torch.Tensor:
    ctx = Context.current()
    V = V.contiguous()
    N = indptr.numel() - 1
    r = V.shape[1]
    Y = torch.empty((N, r), dtype=torch.float64, device=V.device)
    ctx.check(ctx.lib.rt_csr_spmm(ctx.handle, _ptr(indptr), _ptr(indices), _ptr(data), N, _ptr(V), r, r, _ptr(Y), r),
              "rt_csr_spmm")
    return Y


def project_csr(indptr, indices, data, V: torch.Tensor) -> torch.Tensor:
    """A_N = V^T (A V). rt_project_csr."""
    ctx = Context.current()
    V = V.contiguous()
    N = indptr.numel() - 1
    if V.shape[0] != N:
        raise RomtimeHipError("project_csr: V has the wrong number of rows")
    r = V.shape[1]
    AN = torch.empty((r, r), dtype=torch.float64, device=V.device)
    ctx.check(ctx.lib.rt_project_csr(ctx.handle, _ptr(indptr), _ptr(indices), _ptr(data), N, _ptr(V), r, r, _ptr(AN)),
              "rt_project_csr")
    return AN


def project_csr_batched(indptr, indices, data_batch: torch.Tensor, V: torch.Tensor) -> torch.Tensor:
    """data_batch is (nnz x B) (modes as columns, either memory order) -> (B, r, r)."""
    ctx = Context.current()
    V = V.contiguous()
    data_batch, ld, lay = _layout(data_batch)
    nnz, B = data_batch.shape
    # C-ordered (nnz x B) vectors are read in place: a workgroup's entry loads then touch one 8-byte word per
    # 8 B-byte row, which the L2 absorbs (measured: same kernel time as contiguous vectors, and the transpose pass
    # that used to make them contiguous cost 0.25 ms at 5e5 x 120)
    N = indptr.numel() - 1
    r = V.shape[1]
    AN = torch.empty((B, r, r), dtype=torch.float64, device=V.device)
    ctx.check(ctx.lib.rt_project_csr_batched(ctx.handle, _ptr(indptr), _ptr(indices), _ptr(data_batch), ld, lay, B, N,
                                             _ptr(V), r, r, _ptr(AN)), "rt_project_csr_batched")
    return AN


def dense_solve(K: torch.Tensor, b: torch.Tensor):
    """Solve K x = b (K: [r,r] or [B,r,r]; b: [r] or [B,r]). Returns (x, info). Inputs are not modified."""
    ctx = Context.current()
    single = K.dim() == 2
    Kw = K.reshape(-1, K.shape[-2], K.shape[-1]).contiguous().clone()
    bw = b.reshape(Kw.shape[0], -1).contiguous().clone()
    B, r, _ = Kw.shape
    info = torch.zeros(B, dtype=torch.int32, device=K.device)
    ctx.check(ctx.lib.rt_dense_solve_batched(ctx.handle, _ptr(Kw), _ptr(bw), r, B, _ptr(info)),
              "rt_dense_solve_batched")
    return (bw[0] if single else bw), info


def dense_solve_multi(K: torch.Tensor, B: torch.Tensor):
    """X with K X = B for an r x r matrix (r <= 128) and r x nrhs right-hand sides.  Returns (X, info).
    rt_dense_solve_multi."""
    ctx = Context.current()
    K, B = K.contiguous(), B.contiguous()
    r, nrhs = B.shape
    if tuple(K.shape) != (r, r):
        raise RomtimeHipError("dense_solve_multi: K must be r x r and B r x nrhs")
    X = torch.empty_like(B)
    info = torch.zeros(1, dtype=torch.int32, device=K.device)
    ctx.check(ctx.lib.rt_dense_solve_multi(ctx.handle, _ptr(K), r, _ptr(B), _ptr(X), nrhs, _ptr(info)), "rt_dense_solve_multi")
    return X, info


def pod_enqueue(X: torch.Tensor, k: int, normalize: bool):
    """One single-pass POD enqueued on the ctx stream with ONE host call (rt_pod_enqueue): returns the device tensors
    (Q, lam, status2, colnorm) and a tuple of the work buffers to keep alive until the stream has run them."""
    ctx = Context.current()
    X, ld, lay = _layout(X)
    N, n = X.shape
    dev = X.device
    work = torch.empty(n * n + 2 * n + 2 * n * k, dtype=torch.float64, device=dev)
    G, colnorm, lam = work[: n * n].view(n, n), work[n * n: n * n + n], work[n * n + n: n * n + 2 * n]
    Z, Zs = work[n * n + 2 * n: n * n + 2 * n + n * k].view(n, k), work[n * n + 2 * n + n * k:].view(n, k)
    status2 = torch.empty(2, dtype=torch.int32, device=dev)
    Q = torch.empty((N, k), dtype=torch.float64, device=dev)
    ctx.check(ctx.lib.rt_pod_enqueue(ctx.handle, _ptr(X), N, n, ld, lay, k, int(bool(normalize)), _ptr(G), _ptr(colnorm),
                                     _ptr(lam), _ptr(status2), _ptr(Z), _ptr(Zs), _ptr(Q)), "rt_pod_enqueue")
    return Q, lam, status2, colnorm, (work, X)


def tracked_solve(K: torch.Tensor, b: torch.Tensor, Xinv: torch.Tensor = None):
    """Solve K x = b for a batch ([B,r,r], [B,r]) with the inverse tracked in ``Xinv`` ([B,r,r], carried from call to
    call; None = first call).  Returns (x, info, Xinv).  rt_tracked_solve_batched (r <= 80)."""
    ctx = Context.current()
    Kw = K.reshape(-1, K.shape[-2], K.shape[-1]).contiguous()
    bw = b.reshape(Kw.shape[0], -1).contiguous().clone()
    B, r, _ = Kw.shape
    have_prev = Xinv is not None
    if Xinv is None:
        Xinv = torch.zeros_like(Kw)
    info = torch.zeros(B, dtype=torch.int32, device=K.device)
    ctx.check(ctx.lib.rt_tracked_solve_batched(ctx.handle, _ptr(Kw), _ptr(Xinv), _ptr(bw), r, B, int(have_prev), _ptr(info)),
              "rt_tracked_solve_batched")
    return bw, info, Xinv


P1_KINDS = dict(mass=0, stiffness=1, convection=2, trilinear=3, load=4, load_p2=5)


def p1_local_assembly(kind: str, nx: int, rows, cols, h, coef=None, state=None, ramp=None, poly=None) -> torch.Tensor:
    """Closed-form 1-D P1 operator values at the entries (rows[e], cols[e]) for every state: (n_states, m) table.
    ``h`` (n_states) cell sizes, ``coef`` (n_states) optional factors, ``state`` (n_states, nx + 1) nodal values or
    ``ramp`` (n_states) amplitudes of amp * node / nx for the trilinear / load kinds.  ``load_p2`` (the exact integral
    of degree-2 data, fom/heat.py:119): ``state`` is (n_states, 2 nx + 1) vertex / midpoint values, or ``poly``
    (n_states, 3) the coefficients of a0 + a1 x + a2 x^2 in the physical coordinate.  rt_p1_local_assembly."""
    ctx = Context.current()
    rows = rows if isinstance(rows, torch.Tensor) else to_device_index(rows)
    cols = None if cols is None else (cols if isinstance(cols, torch.Tensor) else to_device_index(cols))
    h = to_device(np.atleast_1d(h) if not isinstance(h, torch.Tensor) else h).contiguous()
    n_states, m = h.numel(), rows.numel()
    coef = None if coef is None else to_device(np.atleast_1d(coef) if not isinstance(coef, torch.Tensor) else coef).contiguous()
    mode, st = 0, None
    if state is not None:
        st = to_device(state).contiguous()
        width = 2 * nx + 1 if kind == "load_p2" else nx + 1
        if tuple(st.shape) != (n_states, width):
            raise RomtimeHipError(f"p1_local_assembly: `state` must be (n_states, {width}) for kind {kind!r}")
        mode = 1
    elif ramp is not None:
        st = to_device(np.atleast_1d(ramp) if not isinstance(ramp, torch.Tensor) else ramp).contiguous()
        mode = 2
    elif poly is not None:
        st = to_device(poly).contiguous()
        if tuple(st.shape) != (n_states, 3):
            raise RomtimeHipError("p1_local_assembly: `poly` must be (n_states, 3)")
        mode = 3
    out = torch.empty((n_states, m), dtype=torch.float64, device=h.device)
    ctx.check(ctx.lib.rt_p1_local_assembly(ctx.handle, P1_KINDS[kind], int(nx), _ptr(rows), _ptr(cols), m, n_states, _ptr(h),
                                           _ptr(coef), mode, _ptr(st), _ptr(out)), "rt_p1_local_assembly")
    return out


def sym_eig_values(G: torch.Tensor, first: int = 0, count: int | None = None):
    """Eigenvalues (descending, device) of the symmetric n x n G; (lam, status).  With ``first`` / ``count`` only
    lam[first:first+count] is computed (the rest of ``lam`` is left unset).  rt_sym_eig_values(_part)."""
    ctx = Context.current()
    n = G.shape[0]
    assert G.is_contiguous() and G.shape == (n, n)
    count = n - first if count is None else count
    lam = torch.empty(n, dtype=torch.float64, device=G.device)
    status = torch.zeros(1, dtype=torch.int32, device=G.device)
    ctx.check(ctx.lib.rt_sym_eig_values_part(ctx.handle, _ptr(G), n, first, count, _ptr(lam), _ptr(status)),
              "rt_sym_eig_values_part")
    return lam, status


def sym_eig_vectors(lam: torch.Tensor, k: int, first: int = 0) -> torch.Tensor:
    """Eigenvectors (n x k) of lam[first:first+k] (descending eigenvalues); directly after sym_eig_values.
    rt_sym_eig_vectors."""
    ctx = Context.current()
    n = lam.numel()
    W = torch.empty((n, k), dtype=torch.float64, device=lam.device)
    ctx.check(ctx.lib.rt_sym_eig_vectors(ctx.handle, n, k, _ptr(lam[first:]), _ptr(W)), "rt_sym_eig_vectors")
    return W


def transpose(src: torch.Tensor) -> torch.Tensor:
    ctx = Context.current()
    src = src.contiguous()
    rows, cols = src.shape
    dst = torch.empty((cols, rows), dtype=torch.float64, device=src.device)
    ctx.check(ctx.lib.rt_transpose(ctx.handle, _ptr(src), rows, cols, cols, _ptr(dst), rows), "rt_transpose")
    return dst


def bench_mfma_f64(iters: int = 20000) -> float:
    ctx = Context.current()
    out = C.c_double()
    ctx.check(ctx.lib.rt_bench_mfma_f64(ctx.handle, iters, C.byref(out)), "rt_bench_mfma_f64")
    return out.value


def bench_copy(nbytes: int = 1 << 30, reps: int = 10) -> float:
    ctx = Context.current()
    src = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda").normal_()
    dst = torch.empty_like(src)
    out = C.c_double()
    ctx.check(ctx.lib.rt_bench_copy(ctx.handle, _ptr(dst), _ptr(src), nbytes, reps, C.byref(out)), "rt_bench_copy")
    return out.value
