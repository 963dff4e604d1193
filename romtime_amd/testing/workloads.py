"""Synthetic workloads of the BASELINE.json configurations, shared by ``bench.py``,
``tools/bench_configs.py`` and the full-size GPU tests so that all three run the same inputs.

Nothing here is part of the reference surface: these are the seeded generators of the inputs the hot path is
measured and checked on.  Device work goes through ``romtime_amd.ops`` (the C ABI)."""
from __future__ import annotations

import numpy as np

from .mock import AffineBurgers


def sine_basis(N: int, r: int, seed: int = 1) -> np.ndarray:
    """Orthonormal N x r basis: the r lowest sine modes plus 1e-3 noise, orthonormalised (host QR)."""
    xs = (np.arange(N) + 0.5) / N
    rng = np.random.RandomState(seed)
    V, _ = np.linalg.qr(np.stack([np.sin((k + 1) * np.pi * xs) for k in range(r)], axis=1)
                        + 1e-3 * rng.standard_normal((N, r)))
    return V


def c5_parameters(n_mu: int):
    return [dict(alpha=0.5 + 0.02 * i, beta=1.0 - 0.01 * i, delta=0.3 + 0.005 * i, omega=7.0 + 0.1 * i)
            for i in range(n_mu)]


def affine_tables(mus, nt: int, dt: float):
    """Coefficient tables of ``AffineBurgers`` for every (step, mu), vectorised:
    term_coef (nt x n_mu x 3) = thetas(mu, t), rhs_coef (nt x n_mu x 2) = phis(mu, t)."""
    ts = dt * np.arange(1, nt + 1)[:, None]
    a = np.array([m["alpha"] for m in mus])[None, :]
    b = np.array([m["beta"] for m in mus])[None, :]
    d = np.array([m["delta"] for m in mus])[None, :]
    w = np.array([m["omega"] for m in mus])[None, :]
    term_coef = np.stack([np.broadcast_to(a, (nt, len(mus))), b * np.sin(w * ts), d * np.cos(w * ts)], axis=-1)
    rhs_coef = np.stack([np.sin(w * ts), d * ts], axis=-1)
    return np.ascontiguousarray(term_coef), np.ascontiguousarray(rhs_coef)


def c5_direct(N=100_000, r=80, n_mu=32, nt=10_000, dt=1e-4, seed=5):
    """Config 5, direct path: the descriptor ``rom_bdf_sweep`` consumes (affine operators on one CSR pattern)."""
    fom = AffineBurgers(N=N, nt=nt, dt=dt, bdf2=True, seed=seed)
    mus = c5_parameters(n_mu)
    term_coef, rhs_coef = affine_tables(mus, nt, dt)
    d = dict(indptr=fom.indptr, indices=fom.indices, mass=fom.mass, terms=np.stack([fom.A0, fom.C0, fom.N0]),
             term_coef=term_coef, tril=fom.T, rhs_terms=fom.f, rhs_coef=rhs_coef, dt=dt, bdf2=True)
    return fom, sine_basis(N, r), mus, d


def c5_hyper_reduced(N=100_000, r=80, n_mu=32, nt=10_000, dt=1e-4, m_lin=40, m_nl=120, m_rhs=20, seed=5):
    """Config 5 through the hyper-reduced path (SURVEY 8d "online step, hyper-reduced path"): every reduced operator
    an (M)DEIM expansion, as ``project_reductors`` leaves them (``PT_U``, ``basis_rom``) plus the tables of local
    entries ``F`` that ``assemble(mu, t, entries=dofs)`` would return at every step.

    The expansions represent the AffineBurgers model exactly (its true reduced operators as the leading modes,
    small random padding modes, coefficient tables rotated by a random orthogonal ``PT_U``), so the direct device
    sweep on the same model is a full-size cross-check.  The projections V^T A_q V run on the device (ops).

    Returns ``(terms, d, V, mus)``: ``terms`` = dict(mass, lin, nl, rhs, dt, bdf2) of host arrays for
    ``hrom_bdf_sweep`` / ``oracle.hrom_solve``; ``d`` the direct-path descriptor."""
    from .. import ops

    fom, V, mus, d = c5_direct(N=N, r=r, n_mu=n_mu, nt=nt, dt=dt, seed=seed)
    rng = np.random.RandomState(1)
    rng.standard_normal((N, r))  # keep the stream where tools/bench_configs.py had it (the basis noise draw)
    Vd = ops.to_device(V)
    ip, ix = ops.to_device_index(d["indptr"]), ops.to_device_index(d["indices"])
    proj = lambda vals: ops.project_csr(ip, ix, ops.to_device(vals), Vd).cpu().numpy().reshape(-1)
    rr_of = np.repeat(np.arange(N), np.diff(d["indptr"]))

    def embed(true_cols, coefs, m):
        """true_cols (r^2 x k) with coefficient table coefs (nt x n_mu x k) -> an m-mode term with a random PT_U."""
        k = true_cols.shape[1]
        basis_rom = np.concatenate([true_cols, 1e-3 * rng.standard_normal((true_cols.shape[0], m - k))], axis=1)
        PT_U, _ = np.linalg.qr(rng.standard_normal((m, m)))
        theta = np.concatenate([coefs, np.zeros(coefs.shape[:-1] + (m - k,))], axis=-1)
        return dict(PT_U=PT_U, basis_rom=basis_rom, F=theta @ PT_U.T)

    ones = np.ones((nt, n_mu, 1))
    mass = embed(proj(d["mass"])[:, None], ones, m_lin)
    lin = [embed(proj(d["terms"][q])[:, None], d["term_coef"][:, :, q:q + 1], m_lin) for q in range(3)]
    # trilinear: V^T diag(V u) T V = sum_k u_k N_k
    Nk = np.stack([proj(V[rr_of, k] * d["tril"]) for k in range(r)], axis=1)               # r^2 x r
    nl_full = embed(Nk, np.zeros((1, 1, r)), m_nl)
    nl = dict(PT_U=nl_full["PT_U"], basis_rom=nl_full["basis_rom"],
              W=nl_full["PT_U"] @ np.concatenate([np.eye(r), np.zeros((m_nl - r, r))], axis=0))
    fN = V.T @ d["rhs_terms"].T                                                              # r x F
    rhs = [embed(fN, d["rhs_coef"], m_rhs)]
    return dict(mass=mass, lin=lin, nl=nl, rhs=rhs, dt=dt, bdf2=True), d, V, mus


def c4_snapshots(N=100_000, n_ops=200, seed=4, device=True):
    """Config 4: 200 operator value vectors on a fixed pentadiagonal pattern (nnz = 5 N - 6), smooth combinations
    of 8 spatial profiles plus a 1e-6 noise floor, row 0 zeroed (deim.py:388-389).  Returns (pattern CSR, S) with
    S (nnz x n_ops) a CUDA tensor (``device``) or ndarray."""
    from scipy.sparse import csr_matrix

    rng = np.random.RandomState(seed)
    offs = [-2, -1, 0, 1, 2]
    rows = np.concatenate([np.arange(max(0, -o), min(N, N - o)) for o in offs])
    cols = np.concatenate([np.arange(max(0, -o), min(N, N - o)) + o for o in offs])
    A = csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(N, N))
    A.sort_indices()
    nnz = A.nnz
    x = np.linspace(0, 1, nnz)
    B = np.stack([np.sin((q + 1) * np.pi * x) * (1 + 0.1 * q) for q in range(8)], axis=1)
    theta = rng.standard_normal((8, n_ops))
    S = B @ theta + 1e-6 * rng.standard_normal((nnz, n_ops))
    S[0, :] = 0.0
    if device:
        import torch

        S = torch.from_numpy(S).cuda()
    return A, S
