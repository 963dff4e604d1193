"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle would need minutes
to hours there): orthonormality and the POD energy identity, idempotence of the DEIM interpolant, linearity
and symmetry of the reduced projection, consistency of the device sweep with the class-surface loop."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from romtime_amd import ops as _ops

    return _ops


def _snapshots(N, n, decay, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    s = torch.from_numpy(10.0 ** (-decay * np.arange(n) / (n - 1))).cuda()
    V0, _ = torch.linalg.qr(torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g))
    Z = torch.randn((N, n), dtype=torch.float64, device="cuda", generator=g) / np.sqrt(N)
    return Z @ (s[:, None] * V0.T)


@pytest.mark.parametrize("N,n,r,decay", [(100_000, 256, 40, 6.0), (1_000_000, 512, 40, 8.0), (200_000, 700, 50, 6.0),
                                         (150_000, 1024, 64, 5.0)])
def test_pod_full_size_properties(ops, N, n, r, decay):
    """Configs 2 and 3: Q^T Q = I, sum sigma^2 = ||X_n||_F^2, and ||X_n - Q Q^T X_n||_F^2 = sum_{i>r} sigma_i^2."""
    from romtime_amd import pod

    X = _snapshots(N, n, decay, seed=N % 97)
    out = pod.pod_device(X, num=r, normalize=True)
    Q, s = out["Q"], out["s"]
    assert Q.shape == (N, r) and s.shape == (n,) and out["r"] == r
    QtQ = ops.gemm_tn(Q, Q).cpu().numpy()
    assert np.abs(QtQ - np.eye(r)).max() < 1e-11
    Xn = X / out["colnorm"][None, :]                      # the normalised snapshots of pod.py:31-33
    fro2 = float((Xn * Xn).sum().item())
    assert abs(np.sum(s ** 2) - fro2) <= 1e-12 * fro2      # trace identity (== n for unit columns)
    assert abs(fro2 - n) <= 1e-9 * n
    C = ops.gemm_tn(Q, Xn)                                 # r x n
    res = Xn - ops.gemm_nn(Q, C)
    res2 = float((res * res).sum().item())
    tail2 = float(np.sum(s[r:] ** 2))
    assert abs(res2 - tail2) <= 1e-9 * fro2, (res2, tail2)
    # sigma_i = ||Q_i^T X_n||_2 for the kept modes
    sig = torch.linalg.norm(C, dim=1).cpu().numpy()
    np.testing.assert_allclose(sig, s[:r], rtol=1e-11)


def test_greedy_full_size_idempotence(ops):
    """Config 4 size (5e5 x 120): the DEIM interpolant reproduces every basis vector (it is a projector onto
    the span) and the selected rows are distinct; margins show no accidental tie."""
    N, m = 500_000, 120
    g = torch.Generator(device="cuda").manual_seed(4)
    Phi, _ = torch.linalg.qr(torch.randn((N, m), dtype=torch.float64, device="cuda", generator=g))
    idx, PT_U, margin = ops.deim_greedy(Phi)
    idx_h = idx.cpu().numpy()
    assert len(set(idx_h.tolist())) == m and idx_h.min() >= 0 and idx_h.max() < N
    torch.testing.assert_close(PT_U, Phi[idx], rtol=0, atol=0)       # exact gather
    assert float(margin.min().item()) > 1e-9
    # interpolate a vector of the span from its values at the selected rows only
    coef = torch.randn(m, dtype=torch.float64, device="cuda", generator=g)
    f = ops.gemm_nn(Phi, coef)
    theta, info = ops.dense_solve(PT_U, f[idx])
    assert int(info.abs().sum().item()) == 0
    rec = ops.gemm_nn(Phi, theta)
    assert float((rec - f).abs().max().item()) <= 1e-9 * float(f.abs().max().item())
    # the greedy is deterministic (bitwise reproducible)
    idx2, _, _ = ops.deim_greedy(Phi, want_margin=False)
    assert torch.equal(idx, idx2)


def test_projection_full_size_linearity_and_symmetry(ops):
    """Configs 4/5 size (N = 1e5, nnz ~ 5e5, r = 80): V^T(aA+bB)V = a V^T A V + b V^T B V, symmetric A gives a
    symmetric A_N, and the batched call equals the single calls."""
    from scipy.sparse import csr_matrix

    N, r = 100_000, 80
    rng = np.random.RandomState(9)
    offs = [-2, -1, 0, 1, 2]
    rows = np.concatenate([np.arange(max(0, -o), min(N, N - o)) for o in offs])
    cols = np.concatenate([np.arange(max(0, -o), min(N, N - o)) + o for o in offs])
    pat = csr_matrix((np.ones(rows.size), (rows, cols)), shape=(N, N))
    pat.sort_indices()
    ip, ix = ops.to_device_index(pat.indptr), ops.to_device_index(pat.indices)
    S = csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(N, N))
    S = (S + S.T).tocsr()
    S.sort_indices()
    assert np.array_equal(S.indptr, pat.indptr) and np.array_equal(S.indices, pat.indices)
    a_vals, b_vals = torch.from_numpy(S.data).cuda(), torch.randn(pat.nnz, dtype=torch.float64, device="cuda")
    V, _ = torch.linalg.qr(torch.randn((N, r), dtype=torch.float64, device="cuda"))
    AN = ops.project_csr(ip, ix, a_vals, V)
    BN = ops.project_csr(ip, ix, b_vals, V)
    scale = float(AN.abs().max().item())
    assert float((AN - AN.T).abs().max().item()) <= 1e-12 * scale
    CN = ops.project_csr(ip, ix, 0.3 * a_vals - 1.7 * b_vals, V)
    assert float((CN - (0.3 * AN - 1.7 * BN)).abs().max().item()) <= 1e-12 * scale
    batch = torch.stack([a_vals, b_vals, 0.3 * a_vals - 1.7 * b_vals], dim=1)   # nnz x 3, C order
    out = ops.project_csr_batched(ip, ix, batch, V)
    for got, ref in zip(out, (AN, BN, CN)):
        assert float((got - ref).abs().max().item()) <= 1e-13 * scale
    # against the unfused route: A V by the SpMM kernel, then V^T (A V)
    ref = ops.gemm_tn(V, ops.csr_spmm(ip, ix, a_vals, V))
    assert float((AN - ref).abs().max().item()) <= 1e-12 * scale


def test_sweep_full_size_consistency(ops):
    """Config 5 size (N = 1e5, r = 80, 32 mu), 6 steps: the device sweep equals the same recurrences driven
    step by step through the public operators (project_csr_batched + dense_solve + gemm_nn)."""
    from romtime_amd.sweep import rom_bdf_sweep
    from romtime_amd.testing.mock import AffineBurgers

    N, r, n_mu, nt = 100_000, 80, 32, 6
    fom = AffineBurgers(N=N, nt=nt, dt=1e-4, bdf2=True, seed=5)
    xs = (np.arange(N) + 0.5) / N
    V, _ = np.linalg.qr(np.stack([np.sin((k + 1) * np.pi * xs) for k in range(r)], axis=1)
                        + 1e-3 * np.random.RandomState(1).standard_normal((N, r)))
    mus = [dict(alpha=0.5 + 0.02 * i, beta=1.0 - 0.01 * i, delta=0.3 + 0.005 * i, omega=7.0 + 0.1 * i) for i in range(n_mu)]
    d = fom.descriptor(mus)
    uN = rom_bdf_sweep(V, d["indptr"], d["indices"], d["mass"], d["terms"], d["term_coef"], d["tril"], d["rhs_terms"],
                       d["rhs_coef"], d["dt"], bdf2=True)
    Vd = ops.to_device(V)
    ip, ix = ops.to_device_index(d["indptr"]), ops.to_device_index(d["indices"])
    mass, terms, tril = (ops.to_device(d[k]) for k in ("mass", "terms", "tril"))
    rr = torch.repeat_interleave(torch.arange(N, device="cuda"), torch.from_numpy(np.diff(d["indptr"])).cuda())
    MN = ops.project_csr(ip, ix, mass, Vd)
    fN = ops.gemm_tn(Vd, ops.to_device(d["rhs_terms"]).T.contiguous())          # r x F
    un = torch.zeros((n_mu, r), dtype=torch.float64, device="cuda")
    um = torch.zeros_like(un)
    uh = torch.zeros((n_mu, N), dtype=torch.float64, device="cuda")
    uhp = torch.zeros_like(uh)
    for step in range(nt):
        bdf = 1.5 if step > 0 else 1.0
        th = ops.to_device(d["term_coef"][step])                                 # n_mu x 3
        ustar = 2.0 * uh - uhp
        kv = bdf * mass[None, :] + d["dt"] * (th @ terms + ustar[:, rr] * tril[None, :])
        KN = ops.project_csr_batched(ip, ix, kv.T, Vd)
        rhs = (2.0 * un - 0.5 * um) @ MN.T + d["dt"] * (ops.to_device(d["rhs_coef"][step]) @ fN.T)
        x, _ = ops.dense_solve(KN, rhs)
        um, un = un, x
        uhp, uh = uh, ops.gemm_nn(Vd, x.T.contiguous()).T.contiguous()
        ref = x
        got = uN[:, step, :]
        assert float((got - ref).abs().max().item()) <= 1e-10 * max(float(ref.abs().max().item()), 1e-300), step
