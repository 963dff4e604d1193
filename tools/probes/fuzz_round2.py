"""Randomised shapes through the kernels that changed in round 2 (greedy, fused projection, tracked solve, tall-skinny
product), each against NumPy / SciPy / the oracle.  python3 tools/probes/fuzz_round2.py [iterations] [seed]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.sparse import csr_matrix, random as sprandom, diags
from romtime_amd import ops
from oracle import romtime_oracle as oracle

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for it in range(iters):
    # --- greedy
    N, m = int(rng.randint(40, 20000)), int(rng.randint(1, 70))
    m = min(m, N)
    Phi, _ = np.linalg.qr(rng.standard_normal((N, m)))
    Phi = np.asfortranarray(Phi) if rng.rand() < 0.5 else np.ascontiguousarray(Phi)
    idx, PT_U, _ = ops.deim_greedy(ops.to_device(Phi), want_margin=False)
    ref, _, _ = oracle.deim_greedy(Phi)
    if list(idx.cpu().numpy()) != list(ref):
        bad += 1; print("GREEDY MISMATCH", N, m, flush=True)
    # --- fused projection
    N, r, B = int(rng.randint(33, 6000)), int(rng.randint(1, 129)), int(rng.randint(1, 5))
    bw = int(rng.randint(0, 30))
    A = sprandom(N, N, density=min(1.0, 6.0 / N), random_state=rng, format="csr") if rng.rand() < 0.25 else None
    if A is None:
        offs = sorted(set([0] + list(rng.randint(-bw, bw + 1, size=rng.randint(1, 8)))))
        A = diags([rng.standard_normal(N - abs(o)) for o in offs], offs, shape=(N, N), format="lil")
        for i in rng.choice(N, size=N // 10, replace=False):
            A[i, max(0, i - rng.randint(0, bw + 1))] = 0.0
        A = csr_matrix(A)
    A.eliminate_zeros(); A.sum_duplicates(); A.sort_indices()
    if A.nnz == 0:
        continue
    V, _ = np.linalg.qr(rng.standard_normal((N, min(r, N))))
    r = V.shape[1]
    vals = A.data[:, None] * (1.0 + rng.rand(B))[None, :]
    arr = np.asfortranarray(vals) if rng.rand() < 0.5 else np.ascontiguousarray(vals)
    ip, ix = ops.to_device_index(A.indptr), ops.to_device_index(A.indices)
    AN = ops.project_csr_batched(ip, ix, ops.to_device(arr), ops.to_device(V)).cpu().numpy()
    ref = V.T @ (A @ V)
    for b in range(B):
        want = ref * (vals[0, b] / A.data[0]) if A.data[0] != 0 else None
        want = V.T @ (csr_matrix((vals[:, b], A.indices, A.indptr), shape=(N, N)) @ V)
        err = np.abs(AN[b] - want).max() / max(np.abs(want).max(), 1e-300)
        if not err < 1e-12:
            bad += 1; print("PROJECT MISMATCH", N, r, B, bw, err, flush=True)
    # --- tracked solve
    r, Bs = int(rng.randint(1, 81)), int(rng.randint(1, 9))
    K = rng.standard_normal((Bs, r, r)) / np.sqrt(r) + 2.0 * np.eye(r)
    b = rng.standard_normal((Bs, r))
    x, info, X = ops.tracked_solve(ops.to_device(K), ops.to_device(b))
    K2 = K + 1e-3 * rng.standard_normal((Bs, r, r))
    x2, info2, _ = ops.tracked_solve(ops.to_device(K2), ops.to_device(b), X)
    for xx, KK in ((x, K), (x2, K2)):
        want = np.linalg.solve(KK, b[..., None])[..., 0]
        err = np.abs(xx.cpu().numpy() - want).max() / np.abs(want).max()
        if not err < 1e-10:
            bad += 1; print("SOLVE MISMATCH", r, Bs, err, flush=True)
    # --- tall-skinny product
    N, n, k = int(rng.randint(100, 300000)), int(rng.choice([16, 33, 64, 200, 512])), int(rng.randint(1, 129))
    X = rng.standard_normal((N, n)); T = rng.standard_normal((n, k))
    Y = ops.gemm_nn(ops.to_device(X), ops.to_device(T)).cpu().numpy()
    err = np.abs(Y - X @ T).max() / np.abs(X @ T).max()
    if not err < 1e-12:
        bad += 1; print("GEMM_NN MISMATCH", N, n, k, err, flush=True)
    if it % 10 == 9:
        print("iteration", it + 1, "mismatches", bad, "%.0f s" % (time.time() - t0), flush=True)
print("done:", iters, "iterations,", bad, "mismatches")
sys.exit(1 if bad else 0)
