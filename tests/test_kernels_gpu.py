"""Parity of the HIP kernels (through the C ABI) with the CPU oracle / golden vectors."""
import numpy as np
import pytest
import torch
from scipy.sparse import csr_matrix

from oracle import romtime_oracle as oracle

pytestmark = pytest.mark.gpu
EPS = 2.2e-16


@pytest.fixture(scope="module")
def ops():
    from romtime_amd import ops as _ops

    return _ops


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("N,n", [(1, 1), (7, 3), (64, 16), (1000, 64), (777, 33), (4099, 130), (20000, 256), (3000, 512),
                                 (70001, 200), (33000, 384), (50003, 512), (9000, 97), (16400, 640)])
@pytest.mark.parametrize("order", ["C", "F"])
def test_gram(ops, N, n, order):
    rng = np.random.RandomState(N + n)
    X = rng.standard_normal((N, n))
    X = np.asfortranarray(X) if order == "F" else np.ascontiguousarray(X)
    G = ops.gram(ops.to_device(X)).cpu().numpy()
    ref = X.T @ X
    assert _rel(G, ref) < 5e-14
    np.testing.assert_array_equal(G, G.T)  # exactly symmetric by construction


@pytest.mark.parametrize("N,n,order", [(300_000, 200, "C"), (200_000, 500, "C"), (150_000, 384, "F"), (120_000, 1000, "C"),
                                       (260_000, 250, "F"), (200_000, 641, "C"), (131_077, 512, "C")])
def test_gram_long_sets(ops, N, n, order):
    """Long snapshot sets: the one-launch plan of the Gram kernel where its model takes it (n <= 512, n = 1000), two
    launches elsewhere; a last tile column that is not full is a panel shifted to end at column n (even n) or the
    predicated loader (odd n); ragged K tails.  Exactly symmetric, equal to numpy's X^T X."""
    from romtime_amd._lib import Context

    rng = np.random.RandomState(N % 1000 + n)
    Xh = rng.standard_normal((N, n))
    Xh[:, n // 2] *= 1e3                                     # a misplaced column would show
    X = ops.to_device(np.asfortranarray(Xh) if order == "F" else Xh)
    G = ops.gram(X)
    info = Context.current().launch_info()
    assert info["tile"] == (128, 128)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    one_launch = info["grid"] == 2 * cus          # the one-launch kernel always launches every slot of the chip
    assert one_launch == (n in (200, 250, 384, 500, 512, 1000)), (n, info)
    assert torch.equal(G, G.T)
    assert _rel(G.cpu().numpy(), Xh.T @ Xh) < 5e-14
    assert torch.equal(G, ops.gram(X))


@pytest.mark.parametrize("order", ["C", "F"])
def test_gram_pacing_changes_nothing_but_the_timing(ops, order):
    """The snapshot Gram kernel's workgroups pace themselves for L2 sharing (ctx option "gram_pace", default on): a
    hint, never a condition - the same bits with it off, and with two Grams of different contexts and streams
    competing for the CUs (late workgroups leave the pack instead of holding it back)."""
    import threading

    from romtime_amd._lib import Context

    rng = np.random.RandomState(11)
    Xh = rng.standard_normal((60_011, 384))
    X = ops.to_device(np.asfortranarray(Xh) if order == "F" else Xh)
    ctx = Context.current()
    G_on = ops.gram(X)
    assert ctx.launch_info()["tile"] == (128, 128)          # the gram128 route (the paced kernel)
    ctx.set_option("gram_pace", 0)
    try:
        G_off = ops.gram(X)
    finally:
        ctx.set_option("gram_pace", 1)
    assert torch.equal(G_on, G_off)
    assert _rel(G_on.cpu().numpy(), Xh.T @ Xh) < 5e-14
    out = {}

    def worker(tag):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            st.wait_stream(torch.cuda.default_stream())
            out[tag] = [ops.gram(X) for _ in range(6)]       # this thread's own ctx, its own progress words
            st.synchronize()

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
        assert not t.is_alive()
    for tag in range(3):
        for G in out[tag]:
            assert torch.equal(G, G_on)


def test_mfma_layout_asymmetric(ops):
    """A = I-like selector against an asymmetric B catches swapped C/D row/col maps."""
    N, m, n = 64, 48, 80
    A = np.zeros((N, m))
    A[np.arange(m), np.arange(m)] = 1.0
    B = np.arange(N * n, dtype=float).reshape(N, n) + 0.5
    C = ops.gemm_tn(ops.to_device(A), ops.to_device(B)).cpu().numpy()
    np.testing.assert_array_equal(C, B[:m, :])


@pytest.mark.parametrize("N,m,n", [(500, 10, 7), (4096, 80, 80), (10001, 80, 1), (3333, 1, 5), (2048, 96, 200), (5000, 130, 40)])
@pytest.mark.parametrize("oa,ob", [("C", "C"), ("F", "C"), ("C", "F"), ("F", "F")])
def test_gemm_tn(ops, N, m, n, oa, ob):
    rng = np.random.RandomState(N + m + n)
    A = rng.standard_normal((N, m))
    B = rng.standard_normal((N, n))
    A = np.asfortranarray(A) if oa == "F" else A
    B = np.asfortranarray(B) if ob == "F" else B
    C = ops.gemm_tn(ops.to_device(A), ops.to_device(B)).cpu().numpy()
    assert _rel(C, A.T @ B) < 5e-14


@pytest.mark.parametrize("N,m,n,pad", [(70001, 1, 77, 0), (30000, 7, 201, 3), (20000, 16, 512, 0), (16385, 13, 600, 1),
                                       (40000, 17, 130, 0), (300001, 8, 200, 0)])
def test_gemm_tn_few_modes_streaming_kernel(ops, N, m, n, pad):
    """Row-major A (N x m, a slice of a wider basis) against a tall row-major B with m <= 16 takes the streaming kernel
    of the deflation sweep (rank_update.hip: skinny_tn): ragged row ranges, odd widths and leading dimensions (scalar
    loads), more than one column strip (n > 512); m = 17 falls back to the generic GEMM.  Repeatable bit for bit."""
    from romtime_amd._lib import Context

    rng = np.random.RandomState(N + m + n)
    Qh = rng.standard_normal((N, m + 5))
    Bh = rng.standard_normal((N, n + pad))
    A = ops.to_device(Qh)[:, 2:2 + m]
    B = ops.to_device(Bh)[:, :n]
    C = ops.gemm_tn(A, B)
    tile = Context.current().launch_info()["tile"]
    assert (tile[0] == m) == (m <= 16), tile
    ref = Qh[:, 2:2 + m].T @ Bh[:, :n]
    assert _rel(C.cpu().numpy(), ref) < 5e-14
    assert torch.equal(C, ops.gemm_tn(A, B))


@pytest.mark.parametrize("N,n,k", [(100, 8, 3), (5000, 64, 10), (4097, 256, 40), (3000, 33, 1), (2500, 512, 80), (999, 130, 130)])
@pytest.mark.parametrize("order", ["C", "F"])
def test_gemm_nn(ops, N, n, k, order):
    rng = np.random.RandomState(N + n + k)
    X = rng.standard_normal((N, n))
    X = np.asfortranarray(X) if order == "F" else X
    T = rng.standard_normal((n, k))
    Y = ops.gemm_nn(ops.to_device(X), ops.to_device(T)).cpu().numpy()
    assert _rel(Y, X @ T) < 5e-14


@pytest.mark.parametrize("N,n,k", [(20001, 70, 5), (16390, 64, 64), (40000, 97, 33), (65537, 129, 1)])
def test_gemm_nn_tall_skinny_kernel(ops, N, n, k):
    """Row-major X with N >= 64 x #CUs and k <= 64 takes the streaming kernel (tallskinny.hip): ragged row tail,
    contraction length that is no multiple of the 32-column stage, odd output widths, odd leading dimension."""
    from romtime_amd._lib import Context

    rng = np.random.RandomState(N + n + k)
    Xp = rng.standard_normal((N, n + 3))          # leading dimension n + 3 (odd for even n: scalar-load path)
    X = ops.to_device(Xp)[:, :n]
    T = rng.standard_normal((n, k))
    Y = ops.gemm_nn(X, ops.to_device(T)).cpu().numpy()
    assert Context.current().launch_info()["tile"][0] == 64   # the 64-row streaming tile, not the generic GEMM
    assert _rel(Y, Xp[:, :n] @ T) < 5e-14
    Xc = ops.to_device(np.ascontiguousarray(Xp[:, :n]))       # packed rows (16-byte vector loads when n is even)
    assert _rel(ops.gemm_nn(Xc, ops.to_device(T)).cpu().numpy(), Xp[:, :n] @ T) < 5e-14


def test_gram_scale(ops):
    rng = np.random.RandomState(5)
    X = rng.standard_normal((400, 24)) * 10.0 ** rng.uniform(-3, 3, 24)
    G = ops.gram(ops.to_device(X))
    cn, flag = ops.gram_scale(G, True)
    l2 = np.linalg.norm(X, axis=0)
    np.testing.assert_allclose(cn.cpu().numpy(), l2, rtol=1e-14)
    Xn = X / l2
    np.testing.assert_allclose(G.cpu().numpy(), Xn.T @ Xn, rtol=0, atol=1e-14)
    assert int(flag.item()) == 0
    Z = X.copy()
    Z[:, 3] = 0.0
    G = ops.gram(ops.to_device(Z))
    cn, flag = ops.gram_scale(G, True)
    assert int(flag.item()) == 1 and np.isnan(G.cpu().numpy()[3, 3])


def test_greedy_golden(ops, golden_deim):
    g = golden_deim
    for name in g["names"]:
        B = g[f"basis__{name}"]
        for arr in (np.ascontiguousarray(B), np.asfortranarray(B)):
            idx, PT_U, margin = ops.deim_greedy(ops.to_device(arr))
            idx = idx.cpu().numpy()
            ref = g[f"dofs__{name}"]
            mref = g[f"margin__{name}"]
            # bit-exact wherever the step is not a numerical tie; on exact ties the first index wins
            assert list(idx) == list(ref), (name, idx, ref, mref)
            np.testing.assert_array_equal(PT_U.cpu().numpy(), g[f"PT_U__{name}"])
            np.testing.assert_allclose(margin.cpu().numpy(), mref, rtol=1e-6, atol=1e-11)


@pytest.mark.parametrize("N,m", [(5000, 40), (20011, 120), (1500, 7)])
def test_greedy_random(ops, N, m):
    rng = np.random.RandomState(m)
    B, _ = np.linalg.qr(rng.standard_normal((N, m)))
    dofs, PT_U, margin = oracle.deim_greedy(B)
    idx, PT_U_d, margin_d = ops.deim_greedy(ops.to_device(np.asfortranarray(B)))
    assert list(idx.cpu().numpy()) == list(dofs)
    np.testing.assert_array_equal(PT_U_d.cpu().numpy(), PT_U)
    idx2, _, _ = ops.deim_greedy(ops.to_device(np.ascontiguousarray(B)))
    assert list(idx2.cpu().numpy()) == list(dofs)


@pytest.mark.parametrize("N,m", [(7001, 20), (12000, 33)])
def test_greedy_reads_the_basis_where_it_lies(ops, N, m):
    """The basis is never copied: phase A and the block starts read it in the caller's layout - here a column slice of a
    wider row-major buffer (odd leading dimension, unaligned first column: scalar loads) and a slice of a column-major
    buffer with a padded leading dimension."""
    rng = np.random.RandomState(N + m)
    B, _ = np.linalg.qr(rng.standard_normal((N, m)))
    dofs, PT_U, _ = oracle.deim_greedy(B)
    wide = np.zeros((N, m + 7))
    wide[:, 3:3 + m] = B
    idx, PT_U_d, _ = ops.deim_greedy(ops.to_device(wide)[:, 3:3 + m])
    assert list(idx.cpu().numpy()) == list(dofs)
    np.testing.assert_array_equal(PT_U_d.cpu().numpy(), PT_U)
    tall = np.zeros((m + 2, N + 5))
    tall[1:1 + m, 2:2 + N] = B.T
    idx2, PT_U_d2, _ = ops.deim_greedy(ops.to_device(tall)[1:1 + m, 2:2 + N].T)
    assert list(idx2.cpu().numpy()) == list(dofs)
    np.testing.assert_array_equal(PT_U_d2.cpu().numpy(), PT_U)


def test_greedy_more_workgroups_than_the_chip_holds(ops):
    """N = 3e6: a column's kernel has 2930 workgroups, more than one round of resident ones (256 CUs x 8), so late
    workgroups of a launch start after early ones have finished.  Every workgroup finishes the step before for itself
    from what earlier LAUNCHES left (the other parity's partials, the block's t columns), never from what its own
    launch writes - round 2's kernel read both (wrong pivots beyond ~1-2e6 rows).  Also a second stream
    keeps the chip busy meanwhile, which perturbs the order workgroups are dispatched in."""
    N, m = 3_000_000, 16
    rng = np.random.RandomState(16)
    B = rng.standard_normal((N, m)) / np.sqrt(N)
    B[:, 1:] += 0.3 * B[:, :1]          # correlated columns: the residuals differ from the columns themselves
    dofs, PT_U, margin = oracle.deim_greedy(B)
    assert margin.min() > 1e-6          # no near-tie that rounding could legitimately resolve differently
    Bd = ops.to_device(np.asfortranarray(B))
    side = torch.cuda.Stream()
    filler = torch.randn(4096, 4096, device="cuda")
    for trial in range(3):
        with torch.cuda.stream(side):
            for _ in range(20):
                filler @ filler
        idx, PT_U_d, _ = ops.deim_greedy(Bd)
        assert list(idx.cpu().numpy()) == list(dofs), trial
        np.testing.assert_array_equal(PT_U_d.cpu().numpy(), PT_U)
    torch.cuda.synchronize()


def _penta(N, rng):
    offs = [-2, -1, 0, 1, 2]
    rows, cols = [], []
    for o in offs:
        i = np.arange(max(0, -o), min(N, N - o))
        rows.append(i)
        cols.append(i + o)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(N, N))
    A.sort_indices()
    return A


@pytest.mark.parametrize("N,r", [(300, 8), (5000, 80), (2048, 33), (40000, 128), (1234, 1), (3000, 140)])
def test_project_csr(ops, N, r):
    rng = np.random.RandomState(N)
    A = _penta(N, rng)
    V, _ = np.linalg.qr(rng.standard_normal((N, r)))
    ip, ix = ops.to_device_index(A.indptr), ops.to_device_index(A.indices)
    Vd = ops.to_device(V)
    Y = ops.csr_spmm(ip, ix, ops.to_device(A.data), Vd).cpu().numpy()
    assert _rel(Y, A.dot(V)) < 1e-14
    AN = ops.project_csr(ip, ix, ops.to_device(A.data), Vd).cpu().numpy()
    assert _rel(AN, oracle.project_csr(A, V)) < 1e-13


@pytest.mark.parametrize("r", [7, 40, 72, 80, 120])
@pytest.mark.parametrize("pattern", ["ragged_band", "wide", "mixed", "empty_rows"])
def test_project_fused_pattern_variants(ops, pattern, r):
    """The code paths of the fused projection the banded test matrices do not reach: rows of unequal length inside a
    window (predicated gather), patterns no stage of which can be windowed and patterns where a few can not (the
    kernel's `mixed` variant, gathers out of global memory), empty rows, a last stage shorter than 32 rows, r odd /
    not a multiple of 16, value vectors in both memory orders.  Against scipy's product (utils.py:96-113)."""
    rng = np.random.RandomState(len(pattern) * 100 + r)
    N = 1000 + 13
    if pattern == "ragged_band":
        A = _penta(N, rng).tolil()
        drop = rng.rand(N) < 0.3
        for i in np.nonzero(drop)[0]:
            if 2 <= i < N - 2:
                A[i, i - 2] = 0.0
                A[i, i + 1] = 0.0
        A = csr_matrix(A)
    elif pattern == "wide":
        rows = np.repeat(np.arange(N), 6)
        cols = rng.randint(0, N, size=rows.size)
        A = csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(N, N))
    elif pattern == "mixed":
        A = _penta(N, rng).tolil()
        for i in rng.choice(N, size=12, replace=False):
            A[i, (i + N // 2) % N] = 1.5
        A = csr_matrix(A)
    else:
        A = _penta(N, rng).tolil()
        for i in (0, 5, 6, 7, 500, N - 1):
            A[i, :] = 0.0
        A = csr_matrix(A)
    A.eliminate_zeros()
    A.sum_duplicates()
    A.sort_indices()
    V, _ = np.linalg.qr(rng.standard_normal((N, r)))
    ip, ix = ops.to_device_index(A.indptr), ops.to_device_index(A.indices)
    Vd = ops.to_device(V)
    AN = ops.project_csr(ip, ix, ops.to_device(A.data), Vd).cpu().numpy()
    ref = V.T @ (A @ V)
    assert _rel(AN, ref) < 1e-13
    vals = A.data[:, None] * (1.0 + 0.1 * np.arange(3))[None, :]          # three value vectors on the pattern
    for arr in (np.ascontiguousarray(vals), np.asfortranarray(vals)):
        ANb = ops.project_csr_batched(ip, ix, ops.to_device(arr), Vd).cpu().numpy()
        for b in range(3):
            assert _rel(ANb[b], (1.0 + 0.1 * b) * ref) < 1e-13


def test_project_csr_golden_and_batched(ops, golden_deim):
    g = golden_deim
    ip, ix = ops.to_device_index(g["csr_indptr"]), ops.to_device_index(g["csr_indices"])
    V = ops.to_device(g["mdeim_V"])
    AN = ops.project_csr(ip, ix, ops.to_device(g["csr_data"]), V).cpu().numpy()
    np.testing.assert_allclose(AN, g["project_csr"], rtol=0, atol=1e-13)
    # MDEIM.project_basis fixture: (nnz x m) modes -> (r*r x m)
    rows, cols = g["mdeim_rows"], g["mdeim_cols"]
    N = V.shape[0]
    pat = csr_matrix((np.ones(rows.size), (rows, cols)), shape=(N, N))
    pat.sort_indices()
    ipp, ixp = ops.to_device_index(pat.indptr), ops.to_device_index(pat.indices)
    for arr in (np.ascontiguousarray(g["mdeim_basis_fom"]), np.asfortranarray(g["mdeim_basis_fom"])):
        ANb = ops.project_csr_batched(ipp, ixp, ops.to_device(arr), V).cpu().numpy()
        got = ANb.reshape(ANb.shape[0], -1).T
        np.testing.assert_allclose(got, g["mdeim_basis_rom"], rtol=0, atol=1e-13)


@pytest.mark.parametrize("r,B", [(1, 1), (5, 3), (24, 2), (80, 32), (120, 4), (128, 1)])
def test_dense_solve(ops, r, B):
    rng = np.random.RandomState(r)
    K = rng.standard_normal((B, r, r)) + np.eye(r) * 0.1
    b = rng.standard_normal((B, r))
    x, info = ops.dense_solve(ops.to_device(K), ops.to_device(b))
    ref = np.linalg.solve(K, b[..., None])[..., 0]
    res = np.abs(np.einsum("bij,bj->bi", K, x.cpu().numpy()) - b).max()
    assert res < 1e-10
    np.testing.assert_allclose(x.cpu().numpy(), ref, rtol=1e-8, atol=1e-10)
    assert int(info.abs().sum().item()) == 0


def test_dense_solve_singular_flag(ops):
    K = np.zeros((1, 4, 4))
    x, info = ops.dense_solve(ops.to_device(K), ops.to_device(np.ones((1, 4))))
    assert int(info[0].item()) == 2


@pytest.mark.parametrize("r", [5, 40, 80])
def test_tracked_solve_follows_a_moving_matrix(ops, r):
    """rt_tracked_solve_batched: first call from the safe start, then three steps of a slowly moving K from the carried
    inverse; every answer against np.linalg.solve (what rom.py:492's GMRES converges to)."""
    from romtime_amd._lib import Context

    rng = np.random.RandomState(r)
    B = 6
    K0 = rng.standard_normal((B, r, r)) / np.sqrt(r) + 2.0 * np.eye(r)
    dK = rng.standard_normal((B, r, r)) / np.sqrt(r)
    ctx = Context.current()
    before = ctx.sweep_stats()
    Xinv = None
    for step in range(4):
        K = K0 + 1e-3 * step * dK
        b = rng.standard_normal((B, r))
        x, info, Xinv = ops.tracked_solve(ops.to_device(K), ops.to_device(b), Xinv)
        ref = np.linalg.solve(K, b[..., None])[..., 0]
        np.testing.assert_allclose(x.cpu().numpy(), ref, rtol=1e-10, atol=1e-12)
        assert int(info.abs().sum().item()) == 0
    after = ctx.sweep_stats()
    assert after["lu_fallbacks"] == before["lu_fallbacks"]
    assert after["solves"] - before["solves"] == 4 * B


def test_tracked_solve_falls_back_to_lu_in_the_same_kernel(ops):
    """A matrix the Newton-Schulz iteration cannot invert in its iteration budget (condition number 1e17, diagonal so
    that the exact answer is known), one it can, and an exactly singular one: the first is solved by the pivoted LU
    inside the kernel (counter), the last is flagged."""
    from romtime_amd._lib import Context

    r = 12
    rng = np.random.RandomState(0)
    K = np.zeros((3, r, r))
    d_bad = np.ones(r)
    d_bad[-1] = 1e-17
    K[0] = np.diag(d_bad)[::-1]                    # a row permutation of the diagonal: the LU has to pivot
    K[1] = rng.standard_normal((r, r)) + 3.0 * np.eye(r)
    K[2] = np.ones((r, r))                         # rank one
    b = rng.standard_normal((3, r))
    ctx = Context.current()
    before = ctx.sweep_stats()
    x, info, _ = ops.tracked_solve(ops.to_device(K), ops.to_device(b))
    after = ctx.sweep_stats()
    x = x.cpu().numpy()
    np.testing.assert_allclose(x[0], (b[0] / d_bad[::-1])[::-1], rtol=1e-14)
    np.testing.assert_allclose(x[1], np.linalg.solve(K[1], b[1]), rtol=1e-10, atol=1e-12)
    assert info.cpu().tolist()[:2] == [0, 0] and int(info[2].item()) == 2
    assert after["lu_fallbacks"] - before["lu_fallbacks"] == 2


def test_unsupported_sizes_fail_loudly(ops):
    from romtime_amd._lib import RomtimeHipError

    with pytest.raises(RomtimeHipError):
        ops.dense_solve(torch.eye(200, dtype=torch.float64, device="cuda"), torch.ones(200, dtype=torch.float64, device="cuda"))
    with pytest.raises(RomtimeHipError):
        ops.gemm_tn(torch.ones((4, 2), dtype=torch.float64, device="cuda"), torch.ones((5, 2), dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("n", [3, 17, 64, 200, 256, 512, 513, 777, 1024])
@pytest.mark.parametrize("kind", ["decay", "flat", "cluster"])
def test_sym_eig_device(ops, n, kind):
    """Device tridiagonalisation + multisection + inverse iteration against LAPACK."""
    rng = np.random.RandomState(n)
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    if kind == "decay":
        lam = 10.0 ** (-16.0 * np.arange(n) / max(n - 1, 1))
    elif kind == "flat":
        lam = 1.0 + rng.rand(n)
    else:  # repeated leading eigenvalues and an exactly singular tail
        lam = np.r_[np.full(min(4, n), 2.0), 10.0 ** (-np.arange(n - min(4, n)) * 0.5)]
        lam[n // 2:] = 0.0
    lam = np.sort(lam)[::-1]
    G = (V * lam) @ V.T
    G = 0.5 * (G + G.T)
    Gd = ops.to_device(G)
    ld, status = ops.sym_eig_values(Gd)
    k = min(n, 12)
    W = ops.sym_eig_vectors(ld, k).cpu().numpy()
    assert int(status.item()) == 0
    ref = np.linalg.eigvalsh(G)[::-1]
    got = ld.cpu().numpy()
    assert np.abs(got - ref).max() <= 20 * n * 2.2e-16 * abs(ref[0]), np.abs(got - ref).max()
    np.testing.assert_array_equal(G, Gd.cpu().numpy())  # input untouched
    # residuals: invariant-subspace quality of what inverse iteration returns (clusters are repaired by
    # the Rayleigh-Ritz step in pod.py, so test the span: ||G W - W (W^T G W)|| small after orthonormalising)
    Q, _ = np.linalg.qr(W)
    H = Q.T @ G @ Q
    assert np.abs(G @ Q - Q @ H).max() <= 1e-10 * abs(ref[0])
    np.testing.assert_allclose(np.sort(np.linalg.eigvalsh(H))[::-1], ref[:k], rtol=0, atol=1e-11 * abs(ref[0]))


@pytest.mark.parametrize("bdf2", [True, False])
def test_device_sweep_matches_oracle_loop(bdf2):
    """rt_rom_bdf_sweep against the oracle's restatement of rom.py:430-555 (exact dense solver):
    north-star bar 1e-10 rel-L2 on every parameter point's reduced trajectory."""
    from romtime_amd.sweep import rom_bdf_sweep
    from romtime_amd.testing.mock import AffineBurgers

    fom = AffineBurgers(N=3000, nt=40, dt=2e-3, bdf2=bdf2, seed=3)
    rng = np.random.RandomState(0)
    xs = (np.arange(fom.Nh) + 0.5) / fom.Nh
    V, _ = np.linalg.qr(np.stack([np.sin((k + 1) * np.pi * xs) for k in range(24)], axis=1)
                        + 1e-3 * rng.standard_normal((fom.Nh, 24)))
    mus = [dict(alpha=0.5 + 0.2 * i, beta=1.0 - 0.1 * i, delta=0.3 + 0.05 * i, omega=7.0 + i) for i in range(3)]
    d = fom.descriptor(mus)
    uN = rom_bdf_sweep(V, d["indptr"], d["indices"], d["mass"], d["terms"], d["term_coef"], d["tril"], d["rhs_terms"],
                       d["rhs_coef"], d["dt"], bdf2=bdf2).cpu().numpy()
    assert uN.shape == (3, 40, 24)
    for i, mu in enumerate(mus):
        ref_rom, _ = oracle.rom_solve_nonlinear(fom, V, mu, solver=np.linalg.solve)  # r x nt
        rel = np.linalg.norm(uN[i].T - ref_rom) / np.linalg.norm(ref_rom)
        assert rel <= 1e-10, (i, rel)
        assert np.abs(ref_rom).max() > 1e-4  # a non-trivial trajectory


def test_device_sweep_takes_any_number_of_terms_and_points():
    """The step's coefficient table coef[n_mu][Q] is staged in LDS when it is small; a large one (60 points x 150 terms
    = 72 KB, beyond the 64 KB a launch may ask for: round 2's kernel failed with RT_ERR_HIP there) is read
    from memory instead.  Same model both ways: every affine term split into 50 equal parts."""
    from romtime_amd.sweep import rom_bdf_sweep
    from romtime_amd.testing.mock import AffineBurgers

    fom = AffineBurgers(N=1200, nt=12, dt=2e-3, bdf2=True, seed=4)
    xs = (np.arange(fom.Nh) + 0.5) / fom.Nh
    V, _ = np.linalg.qr(np.stack([np.sin((k + 1) * np.pi * xs) for k in range(10)], axis=1))
    mus = [dict(alpha=0.5 + 0.01 * i, beta=1.0 - 0.01 * i, delta=0.3 + 0.005 * i, omega=7.0 + 0.1 * i) for i in range(60)]
    d = fom.descriptor(mus)
    args = lambda terms, coef: (V, d["indptr"], d["indices"], d["mass"], terms, coef, d["tril"], d["rhs_terms"], d["rhs_coef"], d["dt"])
    small = rom_bdf_sweep(*args(d["terms"], d["term_coef"]), bdf2=True).cpu().numpy()
    parts = 50
    terms = np.repeat(d["terms"], parts, axis=0) / parts                 # 150 x nnz
    coef = np.repeat(d["term_coef"], parts, axis=2)                      # nt x 60 x 150
    big = rom_bdf_sweep(*args(terms, coef), bdf2=True).cpu().numpy()
    assert np.linalg.norm(big - small) <= 1e-11 * np.linalg.norm(small)
    ref, _ = oracle.rom_solve_nonlinear(fom, V, mus[7], solver=np.linalg.solve)
    assert np.linalg.norm(big[7].T - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("m,nrhs", [(120, 6400), (37, 257), (128, 1000), (1, 5), (80, 80)])
def test_dense_solve_multi_matches_lapack(ops, m, nrhs):
    """rt_dense_solve_multi: K X = B for many right-hand sides against one matrix (the fold of PT_U into a hyper-reduced
    operator's expansion, romtime_amd/sweep.py) against np.linalg.solve; a matrix that needs its pivoting; an exactly
    singular one is reported."""
    rng = np.random.RandomState(m)
    K = rng.standard_normal((m, m))
    if m > 2:
        K[0, 0] = 0.0                      # the first pivot cannot be the diagonal entry
        K[[1, 2]] = K[[2, 1]]
    B = rng.standard_normal((m, nrhs))
    X, info = ops.dense_solve_multi(ops.to_device(K), ops.to_device(B))
    ref = np.linalg.solve(K, B)
    assert int(info.item()) == 0
    cond = np.linalg.cond(K)
    assert np.abs(X.cpu().numpy() - ref).max() <= 50 * EPS * cond * np.abs(ref).max()
    resid = np.abs(K @ X.cpu().numpy() - B).max()
    assert resid <= 200 * EPS * m * np.abs(K).max() * np.abs(ref).max()
    if m > 2:
        Ks = K.copy()
        Ks[:, 3 % m] = 0.0
        _, info = ops.dense_solve_multi(ops.to_device(Ks), ops.to_device(B))
        assert int(info.item()) != 0


@pytest.mark.parametrize("R,K,N", [(64, 280, 6400), (64, 144, 320), (7, 2, 256), (33, 290, 1000), (64, 1000, 258),
                                     (64, 281, 6400), (16, 280, 6401)])
def test_few_rows_short_contraction_product(ops, R, K, N):
    """ops.gemm_nn routes few-row (<= 64) / wide-output (>= 256) products to expansion_kernel (sweep.hip: the
    hyper-reduced step's [K_N; M_N] = G Z); odd K or N fall back to the generic GEMM.  Both against the host product."""
    rng = np.random.RandomState(R + K)
    G, Z = rng.standard_normal((R, K)), rng.standard_normal((K, N))
    C = ops.gemm_nn(ops.to_device(G), ops.to_device(Z)).cpu().numpy()
    ref = G @ Z
    assert np.abs(C - ref).max() <= 4 * EPS * K * np.abs(G).max() * np.abs(Z).max() * 4
    # a strided output / operand view takes the same route
    Gd = ops.to_device(np.c_[G, rng.standard_normal((R, 6))])[:, :K]
    out = torch.zeros((R, N + 2), dtype=torch.float64, device="cuda")
    ops.gemm_nn(Gd, ops.to_device(Z), out=out[:, :N])
    assert np.abs(out[:, :N].cpu().numpy() - ref).max() <= 4 * EPS * K * np.abs(G).max() * np.abs(Z).max() * 4
    assert float(out[:, N:].abs().max()) == 0.0


def test_hyper_reduced_sweep_synthetic_and_singular_system(ops):
    """rt_hrom_bdf_sweep on random interpolation terms against oracle.hrom_solve, BDF1 and BDF2, r not a multiple of
    16; one parameter point gets an identically zero K_N: its inverse tracking fails, the device-side LU fallback
    reports the singular system, and the other parameter points are unaffected."""
    from romtime_amd.sweep import hrom_bdf_sweep

    rng = np.random.RandomState(3)
    r, nt, n_mu, dt = 11, 9, 3, 1e-2
    spd = lambda: (lambda a: a @ a.T + r * np.eye(r))(rng.standard_normal((r, r)))

    def matrix_term(m, base, wobble):
        # operator(mu, t) = (1 + wobble(mu, t)) * base + small random modes, as an m-mode interpolation expansion
        cols = np.concatenate([base.reshape(-1, 1), 0.05 * rng.standard_normal((r * r, m - 1))], axis=1)
        PT_U, _ = np.linalg.qr(rng.standard_normal((m, m)))
        theta = np.concatenate([1.0 + wobble[..., None], 0.1 * rng.standard_normal((nt, n_mu, m - 1))], axis=-1)
        return dict(PT_U=PT_U, basis_rom=cols, F=theta @ PT_U.T)

    wob = lambda: 0.1 * rng.standard_normal((nt, n_mu))
    for bdf2 in (True, False):
        mass = matrix_term(4, spd(), 0.0 * wob())
        lin = [matrix_term(3, spd(), wob()), matrix_term(5, rng.standard_normal((r, r)), wob())]
        m_nl = 6
        PTn, _ = np.linalg.qr(rng.standard_normal((m_nl, m_nl)))
        nl = dict(PT_U=PTn, basis_rom=0.3 * rng.standard_normal((r * r, m_nl)), W=0.2 * rng.standard_normal((m_nl, r)),
                  C=0.1 * rng.standard_normal((nt, n_mu, m_nl)), S=1.0 + 0.1 * rng.standard_normal((nt, n_mu)))
        PTf, _ = np.linalg.qr(rng.standard_normal((4, 4)))
        rhs = [dict(PT_U=PTf, basis_rom=rng.standard_normal((r, 4)), F=rng.standard_normal((nt, n_mu, 4)))]
        uN = hrom_bdf_sweep(mass, lin, nl, rhs, dt, bdf2=bdf2).cpu().numpy()
        for b in range(n_mu):
            ref = oracle.hrom_solve(mass, lin, nl, rhs, b, r, nt, dt, bdf2)
            assert np.linalg.norm(uN[b].T - ref) <= 1e-10 * np.linalg.norm(ref), (bdf2, b)
        # the same sweep with steps 1 .. nt-1 replayed as a hipGraph (kernels read the step from a device counter):
        # identical launches, identical numbers
        from romtime_amd._lib import Context
        Context.current().set_option("sweep_graph", 1)
        try:
            uG = hrom_bdf_sweep(mass, lin, nl, rhs, dt, bdf2=bdf2).cpu().numpy()
        finally:
            Context.current().set_option("sweep_graph", 0)
        np.testing.assert_array_equal(uG, uN)
        # parameter point 1: every operator coefficient zero -> K_N = 0
        dead = lambda term: dict(term, F=np.where(np.arange(n_mu)[None, :, None] == 1, 0.0, term["F"]))
        nl_dead = dict(nl, C=np.where(np.arange(n_mu)[None, :, None] == 1, 0.0, nl["C"]),
                       S=np.where(np.arange(n_mu)[None, :] == 1, 0.0, nl["S"]))
        uD = hrom_bdf_sweep(dead(mass), [dead(t) for t in lin], nl_dead, rhs, dt, bdf2=bdf2).cpu().numpy()
        assert not np.all(np.isfinite(uD[1]))                      # singular: no finite answer is claimed
        for b in (0, 2):
            np.testing.assert_allclose(uD[b], uN[b], rtol=0, atol=1e-12 * np.abs(uN[b]).max())


def test_sweeps_beyond_the_tracked_solve(ops):
    """r = 96 > 80: the three padded matrices of the inverse tracking do not fit the LDS, both sweeps take their
    other route (right-hand side kernel, pivoted LU, step closed by its own kernels) - same answers as the oracle."""
    from romtime_amd.sweep import hrom_bdf_sweep, rom_bdf_sweep
    from romtime_amd.testing.mock import AffineBurgers

    rng = np.random.RandomState(7)
    r, nt, n_mu, dt = 96, 5, 2, 1e-2
    spd = lambda: (lambda a: a @ a.T + r * np.eye(r))(rng.standard_normal((r, r)))

    def matrix_term(m, base):
        cols = np.concatenate([base.reshape(-1, 1), 0.05 * rng.standard_normal((r * r, m - 1))], axis=1)
        PT_U, _ = np.linalg.qr(rng.standard_normal((m, m)))
        theta = np.concatenate([1.0 + 0.1 * rng.standard_normal((nt, n_mu, 1)), 0.1 * rng.standard_normal((nt, n_mu, m - 1))], axis=-1)
        return dict(PT_U=PT_U, basis_rom=cols, F=theta @ PT_U.T)

    mass, lin = matrix_term(3, spd()), [matrix_term(4, spd())]
    PTn, _ = np.linalg.qr(rng.standard_normal((5, 5)))
    nl = dict(PT_U=PTn, basis_rom=0.3 * rng.standard_normal((r * r, 5)), W=0.2 * rng.standard_normal((5, r)),
              C=0.1 * rng.standard_normal((nt, n_mu, 5)), S=1.0 + 0.1 * rng.standard_normal((nt, n_mu)))
    PTf, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    rhs = [dict(PT_U=PTf, basis_rom=rng.standard_normal((r, 3)), F=rng.standard_normal((nt, n_mu, 3)))]
    uN = hrom_bdf_sweep(mass, lin, nl, rhs, dt, bdf2=True).cpu().numpy()
    for b in range(n_mu):
        ref = oracle.hrom_solve(mass, lin, nl, rhs, b, r, nt, dt, True)
        assert np.linalg.norm(uN[b].T - ref) <= 1e-10 * np.linalg.norm(ref), b

    fom = AffineBurgers(N=3000, nt=6, dt=2e-3, bdf2=True, seed=3)
    xs = (np.arange(fom.Nh) + 0.5) / fom.Nh
    V, _ = np.linalg.qr(np.stack([np.sin((k + 1) * np.pi * xs) for k in range(r)], axis=1) + 1e-3 * rng.standard_normal((fom.Nh, r)))
    mus = [dict(alpha=0.5, beta=1.0, delta=0.3, omega=7.0), dict(alpha=0.7, beta=0.9, delta=0.35, omega=8.0)]
    d = fom.descriptor(mus)
    uD = rom_bdf_sweep(V, d["indptr"], d["indices"], d["mass"], d["terms"], d["term_coef"], d["tril"], d["rhs_terms"],
                       d["rhs_coef"], d["dt"], bdf2=True).cpu().numpy()
    for i, mu in enumerate(mus):
        ref_rom, _ = oracle.rom_solve_nonlinear(fom, V, mu, solver=np.linalg.solve)
        assert np.linalg.norm(uD[i].T - ref_rom) <= 1e-10 * np.linalg.norm(ref_rom), i


@pytest.mark.parametrize("N,n,k", [(5000, 24, 200), (40000, 8, 512), (3001, 33, 70)])
def test_gemm_nn_accumulate(ops, N, n, k):
    """out = beta out + alpha X T in the GEMM's epilogue (rt_gemm_nn_axpby): the deflation update X -= Q (Q^T X)."""
    rng = np.random.RandomState(N + k)
    X, T, Y0 = rng.standard_normal((N, n)), rng.standard_normal((n, k)), rng.standard_normal((N, k))
    Y = ops.to_device(Y0)
    ret = ops.gemm_nn(ops.to_device(X), ops.to_device(T), out=Y, alpha=-1.0, beta=1.0)
    assert ret is Y
    assert _rel(Y.cpu().numpy(), Y0 - X @ T) < 5e-14
    Y = ops.to_device(Y0)
    ops.gemm_nn(ops.to_device(X), ops.to_device(T), out=Y, alpha=0.5, beta=-2.0)
    assert _rel(Y.cpu().numpy(), -2.0 * Y0 + 0.5 * (X @ T)) < 5e-14
    Y = ops.to_device(np.full((N, k), np.nan))          # beta == 0: the output is not read
    ops.gemm_nn(ops.to_device(X), ops.to_device(T), out=Y)
    assert _rel(Y.cpu().numpy(), X @ T) < 5e-14


@pytest.mark.gpu
@pytest.mark.parametrize("decay,expect_ahead", [(1.0, True), (9.0, False)])
def test_pod_work_enqueued_ahead_of_the_eigenvalues(ops, decay, expect_ahead):
    """orth(num=...) enqueues eigenvectors + back-projection before the eigenvalues reach the host; the host
    checks decide afterwards whether that result stands (shallow spectrum) or is dropped (deep spectrum ->
    deflated levels).  Either way the outcome equals the route taken without looking ahead, and dgesvd."""
    from romtime_amd import pod

    rng = np.random.RandomState(11)
    N, n, k = 6000, 96, 12
    U, _ = np.linalg.qr(rng.standard_normal((N, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    X = (U * 10.0 ** (-decay * np.arange(n) / (n - 1) * (n - 1) / (k - 1))) @ V.T   # sigma_k / sigma_1 = 10^-decay
    Xd = ops.to_device(X)
    key = (n, k, True)
    pod._AHEAD_DROPPED.pop(key, None)
    first = pod.pod_device(Xd, num=k, normalize=True)
    assert pod._AHEAD_DROPPED[key] == (not expect_ahead)
    assert first["passes"] == (1 if expect_ahead else "deflate")
    pod._AHEAD_DROPPED[key] = True            # forces the route without look-ahead
    second = pod.pod_device(Xd, num=k, normalize=True)
    np.testing.assert_array_equal(first["s"], second["s"])
    np.testing.assert_allclose(first["Q"].cpu().numpy(), second["Q"].cpu().numpy(), rtol=0, atol=1e-13)
    Qo, so, _ = oracle.orth(X, num=k, normalize=True)
    # the POD bar of tests/test_surface.py: 2e-13 sigma_1 + 8 eps sigma_1^2 / sigma_i
    assert np.all(np.abs(first["s"][:k] - so[:k]) <= 2e-13 * so[0] + 8 * np.finfo(float).eps * so[0] ** 2 / so[:k])
    Q = first["Q"].cpu().numpy()
    assert np.abs(np.abs(np.sum(Q * Qo, axis=0)) - 1.0).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("N,n,k", [(5000, 200, 8), (4099, 131, 1), (3000, 512, 64), (2000, 96, 65), (777, 33, 5)])
@pytest.mark.parametrize("mode", ["in_place", "out_of_place_scaled", "col_major"])
def test_rank_update(ops, N, n, k, mode):
    """Y diag(d) + alpha X T against NumPy: the streaming kernel (k <= 64, row-major) and the GEMM-epilogue route
    (k > 64 or column-major snapshots) give the same update; Y_src stays intact when out of place."""
    rng = np.random.RandomState(N + n + k)
    Y, X, T = rng.standard_normal((N, n)), rng.standard_normal((N, k)), rng.standard_normal((k, n))
    d = rng.uniform(0.5, 2.0, n)
    Xd, Td = ops.to_device(X), ops.to_device(T)
    if mode == "in_place":
        Yd = ops.to_device(Y)
        out = ops.rank_update(Yd, Xd, Td, alpha=-1.0, out=Yd)
        assert out.data_ptr() == Yd.data_ptr()
        ref = Y - X @ T
    elif mode == "out_of_place_scaled":
        Yd = ops.to_device(Y)
        out = ops.rank_update(Yd, Xd, Td, alpha=-1.0, colscale=ops.to_device(d))
        np.testing.assert_array_equal(Yd.cpu().numpy(), Y)
        ref = Y * d[None, :] - X @ T
    else:
        Yd = ops.to_device(np.asfortranarray(Y))
        out = ops.rank_update(Yd, Xd, Td, alpha=0.5)
        np.testing.assert_array_equal(Yd.cpu().numpy(), Y)
        ref = Y + 0.5 * X @ T
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=1e-12 * max(1, k))


@pytest.mark.gpu
def test_eigensolver_hand_off_forms_agree_bitwise(ops):
    """rt_ctx_set_option("eig_one_xcd"): the one-XCD hand-off (plain stores through one L2) and the write-through
    form move the same numbers, so eigenvalues and eigenvectors are bit-identical; unknown options are refused."""
    from romtime_amd._lib import Context, RomtimeHipError

    ctx = Context.current()
    rng = np.random.RandomState(3)
    n = 300
    A = rng.standard_normal((n, n))
    G = ops.to_device(A @ A.T)
    try:
        ctx.set_option("eig_one_xcd", 1)
        lam1, st1 = ops.sym_eig_values(G)
        W1 = ops.sym_eig_vectors(lam1, 12)
        ctx.set_option("eig_one_xcd", 0)
        lam0, st0 = ops.sym_eig_values(G)
        W0 = ops.sym_eig_vectors(lam0, 12)
    finally:
        ctx.set_option("eig_one_xcd", 1)
    assert int(st1) == 0 and int(st0) == 0
    assert torch.equal(lam1, lam0) and torch.equal(W1, W0)
    ref = np.linalg.eigvalsh(A @ A.T)[::-1]
    np.testing.assert_allclose(lam1.cpu().numpy(), ref, rtol=0, atol=1e-12 * ref[0])
    with pytest.raises(RomtimeHipError):
        ctx.set_option("no_such_option", 1)
