"""POD basis by the method of snapshots on the MI355X (stands where rom/pod.py:7-62 stands).

``orth`` keeps the reference's signature, return values (ALL singular values and the energy
curve, not only the kept ones) and truncation precedence ``tol`` > ``num`` > ``DROP_TOLERANCE``.

Algorithm (device work through the C ABI, see include/romtime_hip.h):

  pass 1   G = X^T X                        rt_gram        (FP64 MFMA; the only O(N n^2) step)
           [row-sharded X: all-reduce G over RCCL/xGMI here]
           colnorm = sqrt(diag G), G <- D^-1 G D^-1        rt_gram_scale  (normalize=True, pod.py:31-33)
           G = W L W^T                      rt_sym_eig_values / _vectors (device; host LAPACK for n > 1024)
           sigma = sqrt(L), energy, truncation -> r
           Q = X (D^-1 W_r S_r^-1)          rt_gemm_nn     (back-projection)

  deep spectra (a kept mode with sigma_r/sigma_1 < TWO_PASS_RATIO): deflated levels (_pod_deflated) --
           accept the modes within 1e-2 of the current largest, X <- X - Q (Q^T X) (rt_gemm_tn/_nn),
           Gram + eigensolve again; every level runs on the device.
  passes=2 keeps the alternative rotation + host Jacobi route (rt_host_jacobi_eigh).

One pass reproduces dgesvd's left singular vectors to ~eps (sigma_1/sigma_i)^2; both deep routes bring
that to ~eps sigma_1/sigma_i, dgesvd's own accuracy (measured against a long-double Jacobi SVD;
DESIGN.md "POD accuracy").
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from . import _lib, ops

DROP_TOLERANCE = 1e-7  # pod.py:4 (the reference's docstring says 1e-8; the code is 1e-7)
DEVICE_EIG = True      # small eigenproblem on the device (rt_sym_eig_*); False = host LAPACK
DEVICE_EIG_MAX_N = 1024
RR_GAP = 1e-4           # smallest eigenvalue gap (relative to lam_1) for which inverse iteration is trusted as is
TWO_PASS_RATIO = 1e-2  # one Gram pass: vectors good to ~eps (sigma_1/sigma_i)^2 <= 2e-12 above this ratio


# shapes (n, num, normalize) whose last POD could not use the work enqueued ahead of the eigenvalues (pod_device)
_AHEAD_DROPPED: dict = {}

# stage timings of the most recent pod_device call in profile mode (rt_ctx_set_profile, which bench.py switches
# on for its timed region): pod_device only records events, stage_timings() waits for them and fills this dict
LAST_TIMINGS: dict = {}
_PENDING_TIMINGS: dict = {}


def stage_timings() -> dict:
    """Stage durations (ms, device time between stream events) of the most recent profiled ``pod_device`` call."""
    if _PENDING_TIMINGS:
        pend = dict(_PENDING_TIMINGS)
        _PENDING_TIMINGS.clear()
        ev = pend["ev"]
        ev[3].synchronize()
        LAST_TIMINGS.clear()
        LAST_TIMINGS["gram_kernel_ms"] = pend["gram_kernel_ms"]
        LAST_TIMINGS["gram_allreduce_ms"] = ev[0].elapsed_time(ev[1])
        LAST_TIMINGS["scale_eigvals_ms"] = ev[1].elapsed_time(ev[2])
        LAST_TIMINGS["eigvec_backproject_ms"] = ev[2].elapsed_time(ev[3])
        LAST_TIMINGS["total_ms"] = ev[0].elapsed_time(ev[3])
        LAST_TIMINGS["host_enqueue_ms"] = pend["host_ms"]
        LAST_TIMINGS["eig_on_device"] = pend["eig_on_device"]
        if pend["levels"] is not None:
            LAST_TIMINGS["levels"] = pend["levels"]
    return LAST_TIMINGS


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


_THREADPOOLS = None  # threadpoolctl.ThreadpoolController, built once: scanning the loaded libraries takes ~2 ms
_HOST_DENSE_LOCK = threading.RLock()   # the pool limits are process-wide: one host dense step at a time


class _blas_threads:
    """Cap the BLAS/OpenMP pools around the small host-side dense steps.  On a GPU box one process
    owns a 16-core share of a 256-thread socket: an uncapped OpenBLAS spins 64+ threads, exhausts
    the cgroup's CPU quota and gets the whole process throttled for tens of milliseconds."""

    def __init__(self, n):
        self.n, self.ctx = n, None

    def __enter__(self):
        global _THREADPOOLS
        _HOST_DENSE_LOCK.acquire()
        try:
            if _THREADPOOLS is None:
                from threadpoolctl import ThreadpoolController

                _THREADPOOLS = ThreadpoolController()
            self.ctx = _THREADPOOLS.limit(limits=self.n)
            self.ctx.__enter__()
        except ImportError:
            self.ctx = None

    def __exit__(self, *exc):
        try:
            if self.ctx is not None:
                self.ctx.__exit__(*exc)
        finally:
            _HOST_DENSE_LOCK.release()


def _eigh_desc(G: np.ndarray):
    """Small symmetric eigenproblem of the n x n Gram matrix on the host (LAPACK dsyevd), eigenvalues
    descending (used by the two-pass route, which needs the full rotation)."""
    with _blas_threads(8):
        lam, W = np.linalg.eigh(G)
    return lam[::-1].copy(), W[:, ::-1].copy(order="C")


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


class _SmallEig:
    """Eigen-decomposition of the n x n Gram matrix: all eigenvalues (host array, descending) at once,
    leading eigenvectors on request (device, n x k).  3 <= n <= 1024 runs on the device
    (rt_sym_eig_values / rt_sym_eig_vectors, Rayleigh-Ritz polish on G when kept eigenvalues are
    closer than RR_GAP * lam_1); other sizes use host LAPACK."""

    def __init__(self, G: torch.Tensor, extra=(), group=None, ahead=None):
        """``ahead(self)``: device work the caller wants enqueued BEFORE the eigenvalues reach the host (it may
        use ``raw_vectors``); the eigenvalues then travel on a side stream while that work runs."""
        self.G, self.n = G, G.shape[0]
        self.on_device = bool(G.is_cuda and DEVICE_EIG and 3 <= self.n <= DEVICE_EIG_MAX_N)
        self.group = group if (group is not None and self.on_device and _world(group) > 1) else None
        self._raw = {}
        if self.on_device:
            if self.group is None:
                self.lam_d, status = ops.sym_eig_values(G)
            else:
                # every rank holds the same G: the tridiagonalisation is replicated (bit-identical), the
                # multisection is split over the ranks and the pieces are all-gathered
                first, cnt = _share(self.n, self.group)
                part, status = ops.sym_eig_values(G, first, cnt)
                pieces = _allgather(torch.cat([part[first:first + cnt], status.to(torch.float64)]), self.group)
                self.lam_d = torch.empty(self.n, dtype=torch.float64, device=G.device)
                for r, piece in enumerate(pieces):
                    f, _ = _share(self.n, self.group, rank=r)
                    self.lam_d[f:f + cnt] = piece[:cnt]
                status = torch.stack([piece[cnt] for piece in pieces]).max().reshape(1)
            head = torch.cat([self.lam_d, status.to(torch.float64)] + [e.to(torch.float64).reshape(-1) for e in extra])
            if ahead is None:
                head = head.cpu().numpy()  # the one device->host transfer of the step
            else:
                head = _fetch_beside(head, lambda: ahead(self))
            self.lam, self.extra = head[: self.n], head[self.n + 1:]
            if int(head[self.n]) != 0:
                # A hand-off of the cooperative tridiagonalisation hit its wall-clock bound: some of its workgroups
                # were not resident (another stream or process held CUs).  Nothing computed from it is used.  This
                # ctx (one per thread and device) leaves the one-XCD form for good and the decomposition is redone
                # once in the general form; if that times out as well, this call takes host LAPACK.  Both are
                # reported: warnings here, "eig_timeouts" through rt_ctx_get_counter.
                import warnings

                ctx = _lib.Context.current()
                if ctx.options.get("eig_one_xcd", 1):
                    warnings.warn("romtime_amd: eigensolver hand-off timed out in the one-XCD form; this context "
                                  "now uses the general form (rt_ctx_set_option eig_one_xcd=0)", RuntimeWarning)
                    ctx.set_option("eig_one_xcd", 0)
                    self.__init__(G, extra=extra, group=group, ahead=ahead)
                    return
                warnings.warn("romtime_amd: eigensolver hand-off timed out in the general form; host LAPACK takes "
                              "this eigenproblem", RuntimeWarning)
                self.on_device, self.group = False, None
        if not self.on_device:
            Gh = G.cpu().numpy()
            self.extra = np.concatenate([np.atleast_1d(e.cpu().numpy()).astype(float) for e in extra]) if extra else np.zeros(0)
            self.lam, self.W = _eigh_desc(Gh)

    def raw_vectors(self, k: int) -> torch.Tensor:
        """Inverse-iteration eigenvectors of the k largest eigenvalues as the device computes them (no host data
        needed: usable before the eigenvalues have been fetched).  Cached per k."""
        if k in self._raw:
            return self._raw[k]
        if self.group is None:
            Z = ops.sym_eig_vectors(self.lam_d, k)
        else:  # each rank back-transforms its share of the k vectors
            first, cnt = _share(k, self.group)
            pieces = _allgather(ops.sym_eig_vectors(self.lam_d, cnt, first=first), self.group)
            Z = torch.empty((self.n, k), dtype=torch.float64, device=self.G.device)
            for r, piece in enumerate(pieces):
                f, _ = _share(k, self.group, rank=r)
                Z[:, f:f + cnt] = piece
        self._raw[k] = Z
        return Z

    def well_separated(self, k: int) -> bool:
        """Inverse iteration resolves an eigenvector to ~eps ||G|| / gap: with every gap among the kept
        eigenvalues (and to the first discarded one) above RR_GAP * lam_1 that is <= 2e-12 and the
        vectors are used as they are; closer eigenvalues get a k x k Rayleigh-Ritz step on G (``vectors``)."""
        n, lam = self.n, self.lam
        gaps = lam[:k] - lam[1:k + 1] if k < n else np.r_[lam[:k - 1] - lam[1:k], lam[k - 1]]
        return bool(gaps.min() >= RR_GAP * max(lam[0], 1e-300))

    def vectors(self, k: int) -> torch.Tensor:
        """n x k eigenvectors of the k largest eigenvalues (must be called before any other device
        operator that uses the ctx workspace, see rt_sym_eig_vectors)."""
        from scipy.linalg import eigh as small_eigh

        n, lam = self.n, self.lam
        if not self.on_device:
            return ops.to_device(np.ascontiguousarray(self.W[:, :k]), self.G.device)
        Z = self.raw_vectors(k)
        if self.well_separated(k):
            return Z
        GZ = ops.gemm_nn(self.G, Z)
        HS = torch.cat([ops.gemm_tn(Z, GZ), ops.gemm_tn(Z, Z)], dim=0).cpu().numpy()
        H, S = 0.5 * (HS[:k] + HS[:k].T), 0.5 * (HS[k:] + HS[k:].T)
        with _blas_threads(1):  # k x k: threads only burn the CPU quota
            theta, C = small_eigh(H, S, check_finite=False)
        theta, C = theta[::-1], np.ascontiguousarray(C[:, ::-1])
        if np.abs(theta - lam[:k]).max() > 1e-9 * max(lam[0], 1e-300):
            raise _lib.RomtimeHipError("device eigenvectors failed the Rayleigh-Ritz cross-check")
        return ops.gemm_nn(Z, ops.to_device(C, Z.device))


_PINNED_TLS = threading.local()   # per thread: concurrent PODs (pipeline.PodWorkers) must not share a staging buffer


def _fetch_beside(head: torch.Tensor, enqueue) -> np.ndarray:
    """Bring ``head`` to the host on a side stream while the work ``enqueue()`` puts on the current stream runs:
    the transfer does not queue behind that work, and the host reads the result while the device is busy."""
    dev = head.device
    main = torch.cuda.current_stream(dev)
    key = (dev.index, head.numel())
    cache = getattr(_PINNED_TLS, "buffers", None)
    if cache is None:
        cache = _PINNED_TLS.buffers = {}
    if key not in cache:
        cache[key] = (torch.empty(head.numel(), dtype=torch.float64).pin_memory(), torch.cuda.Stream(dev))
    pinned, side = cache[key]
    ready = torch.cuda.Event()
    ready.record(main)
    head.record_stream(side)
    with torch.cuda.stream(side):
        side.wait_event(ready)
        pinned.copy_(head, non_blocking=True)
        done = torch.cuda.Event()
        done.record(side)
    enqueue()
    done.synchronize()
    return pinned.numpy().copy()


def _world(group):
    import torch.distributed as dist

    return dist.get_world_size(group)


def _share(total, group, rank=None):
    """(first, count) of this rank's slice of ``total`` items: equal counts, the last slices overlap instead of
    being shorter (the overlapping items are computed twice, identically), so all-gather pieces have one shape."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group) if rank is None else rank
    cnt = -(-total // world)
    return min(rank * cnt, total - cnt), cnt


def _allgather(t, group):
    import torch.distributed as dist

    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t.contiguous(), group=group)
    return out


def _allreduce(G, group):
    if group is not None:
        import torch.distributed as dist

        dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)
    return G


def pod_device(X: torch.Tensor, num=None, tol=None, normalize=True, passes=None, group=None, want_vt=False):
    """POD of a device-resident snapshot matrix (N_local x n).  Returns a dict with
    ``Q`` (device, N_local x r, row-major), ``s``, ``energy`` (host, all n), ``VT`` (host r x n
    or None), ``r``, ``passes``.  With ``group`` (a torch.distributed process group) X is this
    rank's row slab and the Gram matrices are summed over the group.

    ``passes``: None = automatic (one Gram pass when every kept mode has sigma_r/sigma_1 >=
    TWO_PASS_RATIO, otherwise deflated levels), 1 = one pass, 2 = rotation + host Jacobi,
    "deflate" = deflated levels."""
    import time

    if X.dim() != 2:
        raise ValueError("snapshots must be a 2-D array")
    n = X.shape[1]
    ctx = _lib.Context.current() if X.is_cuda else None
    prof = bool(ctx is not None and _profiling(ctx))
    t0 = time.perf_counter()
    # profile mode: stage boundaries are events on the stream, resolved by stage_timings() - nothing here waits
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if prof else None
    if prof:
        ev[0].record()
    # G and the row count share one buffer, so a row-sharded run needs a single all-reduce
    Gbuf = torch.empty(n * n + 1, dtype=torch.float64, device=X.device)
    G = ops.gram(X, out=Gbuf[: n * n].view(n, n))
    Gbuf[n * n:].fill_(float(X.shape[0]))
    _allreduce(Gbuf, group)
    if prof:
        ev[1].record()  # the current stream waits for the collective, so this event closes Gram + all-reduce
    colnorm, flag = ops.gram_scale(G, normalize)
    # ``num`` alone fixes the number of modes before any eigenvalue is known (pod.py:51-53): the eigenvectors and
    # the back-projection of a single-pass POD are then enqueued right behind the eigenvalue kernels, and the
    # host reads the eigenvalues (for sigma, the energy curve and the checks below) while they run.  If the
    # checks say the spectrum needs the deflated route or a Rayleigh-Ritz step, that result is dropped.
    k_ahead = int(min(num, n)) if (num and not tol and passes in (None, 1) and X.is_cuda) else 0  # same on every rank
    ahead_key = (n, k_ahead, bool(normalize))
    if _AHEAD_DROPPED.get(ahead_key):
        k_ahead = 0  # the last POD of this shape needed the deflated route: do not enqueue work that is thrown away
    ahead_out = {}

    def ahead(e):
        if prof:
            ev[2].record()  # eigenvalues enqueued; what follows is eigenvectors + back-projection
        if not e.on_device or k_ahead < 1:
            return
        Z = e.raw_vectors(k_ahead)
        Zs = ops.backproject_weights(Z, e.lam_d, colnorm if normalize else None)   # D^-1 W S^-1 in one launch
        ahead_out["Q"] = ops.gemm_nn(X, Zs)

    eig = _SmallEig(G, extra=(flag, Gbuf[n * n:]), group=group, ahead=ahead if k_ahead else None)
    if prof and not k_ahead:
        ev[2].record()
    if normalize and int(eig.extra[0]) != 0:
        # the reference divides by a zero norm and scipy.linalg.svd then rejects the NaNs (pod.py:32-38)
        raise ValueError("array must not contain infs or NaNs (zero-norm snapshot with normalize=True)")
    lam = eig.lam
    s = np.sqrt(np.clip(lam, 0.0, None))
    energy = _energy(s)
    r = truncation_rank(s, energy, num=num, tol=tol)
    deep = r > 0 and s[0] > 0 and s[r - 1] < TWO_PASS_RATIO * s[0]
    if passes is None:
        passes = "deflate" if deep else 1
    _AHEAD_DROPPED[ahead_key] = bool(passes != 1 or r != ahead_key[1])

    if passes == 1:
        if r > 0 and r == k_ahead and "Q" in ahead_out and eig.on_device and eig.well_separated(r):
            Q = ahead_out["Q"]  # enqueued before the eigenvalues arrived, and they confirm it
            VT = np.ascontiguousarray(eig.raw_vectors(r).cpu().numpy().T) if want_vt else None
        elif r > 0:
            Z = eig.vectors(r)
            Zs = (Z / colnorm[:, None] if normalize else Z) * ops.to_device(_inv_or_zero(s[:r]), X.device)[None, :]
            Q = ops.gemm_nn(X, Zs.contiguous())
            VT = np.ascontiguousarray(Z.cpu().numpy().T) if want_vt else None
        else:
            Q, VT = X.new_zeros((X.shape[0], 0)), (np.zeros((0, n)) if want_vt else None)
    elif passes == 2:
        W = eig.vectors(n)
        Y = ops.gemm_nn(X, (W / colnorm[:, None] if normalize else W).contiguous())
        G2 = _allreduce(ops.gram(Y), group).cpu().numpy()
        lam2, W2 = jacobi_eigh(G2)
        s = np.sqrt(np.clip(lam2, 0.0, None))
        energy = _energy(s)
        r = truncation_rank(s, energy, num=num, tol=tol)
        T2 = W2[:, :r] * _inv_or_zero(s[:r])
        Q = ops.gemm_nn(Y, ops.to_device(T2, X.device)) if r > 0 else X.new_zeros((X.shape[0], 0))
        VT = np.ascontiguousarray((W.cpu().numpy() @ W2[:, :r]).T) if want_vt else None
    elif passes == "deflate":
        Q, s, energy, r, VT, levels = _pod_deflated(X, eig, colnorm, normalize, num, tol, group, want_vt)
    else:
        raise ValueError(f"passes must be None, 1, 2 or 'deflate', not {passes!r}")
    # more snapshots than DoFs: the thin SVD of the reference has only min(N, n) singular values (pod.py:38)
    n_rows = int(round(float(eig.extra[1])))  # global row count (summed with G over the ranks)
    if n_rows < n:
        s, energy, r = s[:n_rows], energy[:n_rows], min(r, n_rows)
        Q = Q[:, :r].contiguous()
        VT = VT[:r] if VT is not None else None
    if prof:
        ev[3].record()
        _PENDING_TIMINGS.clear()
        _PENDING_TIMINGS.update(ev=ev, gram_kernel_ms=ctx.last_gram_ms(), host_ms=1e3 * (time.perf_counter() - t0),
                                eig_on_device=float(eig.on_device),
                                levels=float(levels) if passes == "deflate" else None)
    out = dict(Q=Q, s=s, energy=energy, VT=VT, r=r, passes=passes, colnorm=colnorm)
    if prof:
        out["gram_kernel_ms"] = _PENDING_TIMINGS["gram_kernel_ms"]
    return out


MAX_LEVELS = 12


def _pod_deflated(X, eig0, colnorm, normalize, num, tol, group, want_vt):
    """Deep spectra without leaving the device: a Gram pass resolves the modes within TWO_PASS_RATIO of
    the current largest singular value to ~eps; those are accepted, projected out of the snapshots
    (X <- X - Q (Q^T X), an O(eps ||X||) perturbation - what dgesvd's backward stability allows too) and the
    next level starts from a matrix whose largest singular value is >= 100x smaller.  Each level costs a
    Gram pass, one small eigensolve and two tall-skinny GEMMs; modes come out with errors ~eps sigma_1 /
    (sigma_i gap), dgesvd's own level (POD accuracy)."""
    n = X.shape[1]
    dev = X.device
    total = float(np.sum(np.clip(eig0.lam, 0.0, None)))   # == trace(G0): the energy denominator
    Xc, eig = None, eig0
    s_acc, Q_acc, W_acc = [], [], []
    # with ``num`` alone the basis never has more than num columns: the levels write straight into it
    cap = int(min(num, n)) if (num and not tol) else None
    # zeros: with ``num`` above the numerical rank and an exactly zero tail the levels stop before the buffer is full,
    # and the columns never written must be the zero columns the single-pass route returns (_inv_or_zero)
    Qbuf = torch.zeros((X.shape[0], cap), dtype=torch.float64, device=dev) if cap else None
    levels = 0
    while True:
        levels += 1
        sig = np.sqrt(np.clip(eig.lam, 0.0, None))
        have = sum(len(x) for x in s_acc)
        room = (cap if cap else n) - have
        # below n eps sigma_1 the deflated snapshots hold rounding residue, not modes: the numerical rank is reached
        # and the remaining columns of a ``num`` basis stay zero (what the single-pass route returns too)
        floor = n * np.finfo(float).eps * s_acc[0][0] if s_acc else 0.0
        k = int(min(max(1, np.count_nonzero(sig >= TWO_PASS_RATIO * sig[0])), room)) if sig[0] > floor else 0
        if k > 0:
            Z = eig.vectors(k)
            src = X if Xc is None else Xc
            scale = ops.to_device(_inv_or_zero(sig[:k]), dev)[None, :]
            Zs = (Z / colnorm[:, None] if (normalize and Xc is None) else Z) * scale
            Ql = ops.gemm_nn(src, Zs.contiguous(), out=Qbuf[:, have:have + k] if cap else None)
            s_acc.append(sig[:k])
            Q_acc.append(Ql)
            if want_vt:
                W_acc.append(Z.cpu().numpy())
        got = sum(len(x) for x in s_acc)
        tail = sig[k:k + (n - got)]
        s_full = np.concatenate(s_acc + [tail, np.zeros(max(0, n - got - len(tail)))])
        ev = np.power(s_full, 2)
        energy = np.cumsum(ev) / total
        r = truncation_rank(s_full, energy, num=num, tol=tol)
        if r <= got or k == 0 or got >= n or levels >= MAX_LEVELS or tail.size == 0 or tail[0] <= 0.0:
            break
        # deflate: X <- X - Q (Q^T X), twice (classical Gram-Schmidt needs the second sweep; taking its
        # coefficients from the k x k matrix Q^T Q instead, C = (2I - Q^T Q) Q^T X, saves a pass over the snapshots
        # but leaves the rounding of the first update along Q in place: deep modes 2.5x less accurate).
        # The first sweep of the first level reads the caller's X (which must stay intact) and writes the working
        # copy, folding the column normalisation in; every later sweep updates the working copy in place
        # (rt_rank_update).
        for sweep in range(2):
            if Xc is None:
                inv = (1.0 / colnorm) if normalize else None
                C = _allreduce(ops.gemm_tn(Ql, X), group)   # Q^T X sums over the row slabs of all ranks
                if inv is not None:
                    C = C * inv[None, :]
                Xc = ops.rank_update(X, Ql, C, alpha=-1.0, colscale=inv)
            else:
                C = _allreduce(ops.gemm_tn(Ql, Xc), group)
                ops.rank_update(Xc, Ql, C, alpha=-1.0, out=Xc)
        G = _allreduce(ops.gram(Xc), group)
        eig = _SmallEig(G, group=group)
    if cap:
        Q = Qbuf if r == cap else Qbuf[:, :r].contiguous()
    else:
        Q = torch.cat(Q_acc, dim=1)[:, :r].contiguous() if (r > 0 and Q_acc) else X.new_zeros((X.shape[0], 0))
    VT = np.ascontiguousarray(np.hstack(W_acc)[:, :r].T) if (want_vt and W_acc) else (np.zeros((0, n)) if want_vt else None)
    return Q, s_full, energy, r, VT, levels


def orth(snapshots, num=None, tol=None, normalize=True, return_VT=False, passes=None, group=None):
    """Drop-in for ``romtime.rom.pod.orth`` (pod.py:7-62).

    ndarray in -> ndarrays out ``(Q, s, energy[, VT])``; a float64 CUDA tensor in -> ``Q`` stays on
    the device (``s``, ``energy``, ``VT`` are small host arrays).  Columns of ``Q`` are defined up
    to sign, as with any SVD.  ``passes`` (None = automatic, 1, 2) and ``group`` are extensions.

    Which entries of ``s`` to trust.  ALL singular values are returned, as the reference does (its reports store
    them, rom.py:338-340,386-388).  They come from Gram matrices, so an entry carries an ABSOLUTE error of about
    eps sigma_L^2 / sigma_i, sigma_L being the largest singular value of the level that produced it: every kept
    mode, and every mode the truncation rule had to look at, is resolved to dgesvd's accuracy (deflated levels restart
    the scale at each level); entries further down the tail than that - below sqrt(eps) sigma_L of the last level and
    not needed to decide the rank - are accurate to that absolute bound only (an exactly zero singular value can read
    1e-8 sigma_1).  ``energy`` is insensitive to this (it sums sigma^2)."""
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
