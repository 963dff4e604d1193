"""POD basis by the method of snapshots on the MI355X (stands where rom/pod.py:7-62 stands).

``orth`` keeps the reference's signature, return values (ALL singular values and the energy
curve, not only the kept ones) and truncation precedence ``tol`` > ``num`` > ``DROP_TOLERANCE``.

Algorithm (device work through the C ABI, see include/romtime_hip.h):

  pass 1   G = X^T X                        rt_gram        (FP64 MFMA; the only O(N n^2) step)
           [row-sharded X: all-reduce G over RCCL/xGMI here]
           colnorm = sqrt(diag G), G <- D^-1 G D^-1        rt_gram_scale  (normalize=True, pod.py:31-33)
           G = W L W^T                      small n x n symmetric eigenproblem (host LAPACK)
           sigma = sqrt(L), energy, truncation -> r
           Q = X (D^-1 W_r S_r^-1)          rt_gemm_nn     (back-projection)

  pass 2   (only when a kept mode has sigma_r/sigma_1 < TWO_PASS_RATIO, or passes=2)
           Y = X (D^-1 W)   full rotation   rt_gemm_nn
           G2 = Y^T Y                       rt_gram  [+ all-reduce]
           G2 = W2 L2 W2^T                  rt_host_jacobi_eigh (relative accuracy on the graded G2)
           Q = Y (W2_r S_r^-1)              rt_gemm_nn

One pass reproduces dgesvd's left singular vectors to ~eps (sigma_1/sigma_i)^2; the second
pass brings that to ~eps sigma_1/sigma_i, which is dgesvd's own accuracy (measured against a
long-double Jacobi SVD; DESIGN.md "POD accuracy").
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, ops

DROP_TOLERANCE = 1e-7  # pod.py:4 (the reference's docstring says 1e-8; the code is 1e-7)
DEVICE_EIG = True      # small eigenproblem on the device (rt_sym_eig_*); False = host LAPACK
DEVICE_EIG_MAX_N = 512
RR_GAP = 1e-4           # smallest eigenvalue gap (relative to lam_1) for which inverse iteration is trusted as is
TWO_PASS_RATIO = 1e-2  # one Gram pass: vectors good to ~eps (sigma_1/sigma_i)^2 <= 2e-12 above this ratio


# stage timings of the most recent pod_device call (host wall clock, ms); filled only when the
# ctx is in profile mode (rt_ctx_set_profile), which bench.py switches on for its timed region
LAST_TIMINGS: dict = {}


def truncation_rank(s, energy, num=None, tol=None) -> int:
    """Number of modes kept, with the reference's precedence (pod.py:46-57).

    ``energy < tol`` is strict, so the mode that crosses ``tol`` is excluded; the energy curve is
    non-decreasing and ``s`` non-increasing, hence every mask is a prefix."""
    if tol:
        return int(np.count_nonzero(energy < tol))
    if num:
        return int(min(num, len(s)))
    return int(np.count_nonzero(s > DROP_TOLERANCE))


def _profiling(ctx) -> bool:
    return bool(getattr(ctx, "profiling", False))


class _blas_threads:
    """Cap the BLAS/OpenMP pools around the small host-side dense steps.  On a GPU box one process
    owns a 16-core share of a 256-thread socket: an uncapped OpenBLAS spins 64+ threads, exhausts
    the cgroup's CPU quota and gets the whole process throttled for tens of milliseconds."""

    def __init__(self, n):
        self.n, self.ctx = n, None

    def __enter__(self):
        try:
            from threadpoolctl import threadpool_limits

            self.ctx = threadpool_limits(limits=self.n)
            self.ctx.__enter__()
        except ImportError:
            self.ctx = None

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def _eigh_desc(G: np.ndarray):
    """Small symmetric eigenproblem of the n x n Gram matrix on the host (LAPACK dsyevd), eigenvalues
    descending (used by the two-pass route, which needs the full rotation)."""
    with _blas_threads(8):
        lam, W = np.linalg.eigh(G)
    return lam[::-1].copy(), np.ascontiguousarray(W[:, ::-1])


def jacobi_eigh(G: np.ndarray, max_sweeps: int = 40):
    """Host two-sided Jacobi (rt_host_jacobi_eigh): eigenvalues descending, eigenvectors in columns."""
    lib = _lib.load()
    n = G.shape[0]
    A = np.array(G, dtype=np.float64, order="C", copy=True)
    W = np.empty((n, n))
    lam = np.empty(n)
    sweeps = C.c_int(0)
    rc = lib.rt_host_jacobi_eigh(A.ctypes.data, n, W.ctypes.data, lam.ctypes.data, max_sweeps, C.byref(sweeps))
    if rc != 0:
        raise _lib.RomtimeHipError(f"rt_host_jacobi_eigh failed ({rc})")
    return lam, W


def _energy(s):
    ev = np.power(s, 2)
    return np.cumsum(ev) / np.sum(ev)


def _inv_or_zero(s):
    out = np.zeros_like(s)
    nz = s > 0
    out[nz] = 1.0 / s[nz]
    return out


def pod_device(X: torch.Tensor, num=None, tol=None, normalize=True, passes=None, group=None, want_vt=False):
    """POD of a device-resident snapshot matrix (N_local x n).  Returns a dict with
    ``Q`` (device, N_local x r, row-major), ``s``, ``energy`` (host, all n), ``VT`` (host r x n
    or None), ``r``, ``passes``.  With ``group`` (a torch.distributed process group) X is this
    rank's row slab and the Gram matrices are summed over the group."""
    if X.dim() != 2:
        raise ValueError("snapshots must be a 2-D array")
    n = X.shape[1]

    def allreduce(G):
        if group is not None:
            import torch.distributed as dist

            dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)
        return G

    import time

    ctx = _lib.Context.current() if X.is_cuda else None
    prof = bool(ctx is not None and ctx.lib is not None and _profiling(ctx))
    t0 = time.perf_counter()
    G = ops.gram(X)
    if prof:
        LAST_TIMINGS.clear()
        LAST_TIMINGS["gram_kernel_ms"] = ctx.last_gemm_ms()
        LAST_TIMINGS["gram_ms"] = 1e3 * (time.perf_counter() - t0)
    G = allreduce(G)
    colnorm, flag = ops.gram_scale(G, normalize)
    dev_eig = X.is_cuda and passes != 2 and 3 <= n <= DEVICE_EIG_MAX_N and DEVICE_EIG
    if dev_eig:
        out = _pod_device_eig(X, G, colnorm, flag, normalize, num, tol, passes, want_vt, prof, t0)
        if out is not None:
            return out
    Gh = G.cpu().numpy()
    t1 = time.perf_counter()
    if normalize and int(flag.item()) != 0:
        # the reference divides by a zero norm and scipy.linalg.svd then rejects the NaNs (pod.py:32-38)
        raise ValueError("array must not contain infs or NaNs (zero-norm snapshot with normalize=True)")
    lam, W = _eigh_desc(Gh)
    t2 = time.perf_counter()
    s = np.sqrt(np.clip(lam, 0.0, None))
    energy = _energy(s)
    r = truncation_rank(s, energy, num=num, tol=tol)
    dinv = (1.0 / colnorm.cpu().numpy()) if normalize else None

    if passes is None:
        passes = 2 if (r > 0 and s[0] > 0 and s[r - 1] < TWO_PASS_RATIO * s[0]) else 1

    if passes == 1:
        T = W[:, :r] * _inv_or_zero(s[:r])
        if dinv is not None:
            T = T * dinv[:, None]
        Q = ops.gemm_nn(X, ops.to_device(T, X.device)) if r > 0 else X.new_zeros((X.shape[0], 0))
        VT = np.ascontiguousarray(W[:, :r].T) if want_vt else None
    else:
        T1 = W if dinv is None else W * dinv[:, None]
        Y = ops.gemm_nn(X, ops.to_device(T1, X.device))
        G2 = allreduce(ops.gram(Y)).cpu().numpy()
        lam2, W2 = jacobi_eigh(G2)
        s = np.sqrt(np.clip(lam2, 0.0, None))
        energy = _energy(s)
        r = truncation_rank(s, energy, num=num, tol=tol)
        T2 = W2[:, :r] * _inv_or_zero(s[:r])
        Q = ops.gemm_nn(Y, ops.to_device(T2, X.device)) if r > 0 else X.new_zeros((X.shape[0], 0))
        VT = np.ascontiguousarray((W @ W2[:, :r]).T) if want_vt else None
    if prof:
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        LAST_TIMINGS["allreduce_scale_d2h_ms"] = 1e3 * (t1 - t0) - LAST_TIMINGS["gram_ms"]
        LAST_TIMINGS["eig_host_ms"] = 1e3 * (t2 - t1)
        LAST_TIMINGS["backproject_ms"] = 1e3 * (t3 - t2)
        LAST_TIMINGS["total_ms"] = 1e3 * (t3 - t0)
    return dict(Q=Q, s=s, energy=energy, VT=VT, r=r, passes=passes, colnorm=colnorm)


def _pod_device_eig(X, G, colnorm, flag, normalize, num, tol, passes, want_vt, prof, t0):
    """Single-pass POD with the n x n eigenproblem on the device (rt_sym_eig_values/_vectors) and a
    k x k Rayleigh-Ritz polish.  Returns None when the spectrum asks for the two-pass path (the caller
    continues on the host-eig route, which provides the full rotation)."""
    import time

    from scipy.linalg import eigh as small_eigh

    n = G.shape[0]
    lam_d, status = ops.sym_eig_values(G)
    head = torch.cat([lam_d, status.to(torch.float64), flag.to(torch.float64)]).cpu().numpy()  # one D2H
    t1 = time.perf_counter()
    lam, eig_status, zero_norm = head[:n], int(head[n]), int(head[n + 1])
    if normalize and zero_norm != 0:
        raise ValueError("array must not contain infs or NaNs (zero-norm snapshot with normalize=True)")
    if eig_status != 0:
        raise _lib.RomtimeHipError("rt_sym_eig_values: inter-workgroup hand-off timed out")
    s = np.sqrt(np.clip(lam, 0.0, None))
    energy = _energy(s)
    r = truncation_rank(s, energy, num=num, tol=tol)
    if passes is None and r > 0 and s[0] > 0 and s[r - 1] < TWO_PASS_RATIO * s[0]:
        return None
    if r == 0:
        Q = X.new_zeros((X.shape[0], 0))
        VT = np.zeros((0, n)) if want_vt else None
    else:
        Z = ops.sym_eig_vectors(lam_d, r)                       # n x r
        # inverse iteration resolves an eigenvector to ~eps ||G|| / gap: with every gap among the kept
        # eigenvalues (and to the first discarded one) above RR_GAP * lam_1 that is <= 2e-12 and the
        # vectors are used as they are; closer eigenvalues get the k x k Rayleigh-Ritz step on G
        gaps = lam[:r] - lam[1:r + 1] if r < n else np.r_[lam[:r - 1] - lam[1:r], lam[r - 1]]
        if gaps.min() >= RR_GAP * max(lam[0], 1e-300):
            C, C_is_identity = None, True
        else:
            GZ = ops.gemm_nn(G, Z)
            HS = torch.cat([ops.gemm_tn(Z, GZ), ops.gemm_tn(Z, Z)], dim=0).cpu().numpy()
            H, S = 0.5 * (HS[:r] + HS[:r].T), 0.5 * (HS[r:] + HS[r:].T)
            with _blas_threads(1):                                   # k x k: threads only burn the CPU quota
                theta, C = small_eigh(H, S)                          # generalised Rayleigh-Ritz
            theta, C = theta[::-1], C[:, ::-1]
            if np.abs(theta - lam[:r]).max() > 1e-9 * max(lam[0], 1e-300):
                raise _lib.RomtimeHipError("device eigenvectors failed the Rayleigh-Ritz cross-check")
            C_is_identity = False
        scale = _inv_or_zero(s[:r])
        T2 = ops.to_device(np.ascontiguousarray(np.diag(scale) if C_is_identity else C * scale), X.device)
        Zs = Z if not normalize else Z / colnorm[:, None]
        if C_is_identity:
            Q = ops.gemm_nn(X, (Zs * T2.diagonal()[None, :]).contiguous())
        else:
            Q = ops.gemm_nn(X, ops.gemm_nn(Zs.contiguous(), T2))
        if want_vt:
            Zh = Z.cpu().numpy()
            VT = np.ascontiguousarray((Zh if C_is_identity else Zh @ C).T)
        else:
            VT = None
    if prof:
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        LAST_TIMINGS["allreduce_scale_eigvals_ms"] = 1e3 * (t1 - t0) - LAST_TIMINGS["gram_ms"]
        LAST_TIMINGS["eigvec_backproject_ms"] = 1e3 * (t3 - t1)
        LAST_TIMINGS["total_ms"] = 1e3 * (t3 - t0)
        LAST_TIMINGS["eig_path"] = 1.0
    return dict(Q=Q, s=s, energy=energy, VT=VT, r=r, passes=1, colnorm=colnorm)


def orth(snapshots, num=None, tol=None, normalize=True, return_VT=False, passes=None, group=None):
    """Drop-in for ``romtime.rom.pod.orth`` (pod.py:7-62).

    ndarray in -> ndarrays out ``(Q, s, energy[, VT])``; a float64 CUDA tensor in -> ``Q`` stays on
    the device (``s``, ``energy``, ``VT`` are small host arrays).  Columns of ``Q`` are defined up
    to sign, as with any SVD.  ``passes`` (None = automatic, 1, 2) and ``group`` are extensions."""
    if isinstance(snapshots, list):
        raise ValueError("You should use an array, not a list.")
    on_host = not isinstance(snapshots, torch.Tensor)
    X = ops.to_device(snapshots)
    out = pod_device(X, num=num, tol=tol, normalize=(normalize == True), passes=passes, group=group,  # noqa: E712
                     want_vt=return_VT)
    Q = out["Q"].cpu().numpy() if on_host else out["Q"]
    if return_VT:
        return Q, out["s"], out["energy"], out["VT"]
    return Q, out["s"], out["energy"]
