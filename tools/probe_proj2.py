import sys, numpy as np, torch
from scipy.sparse import csr_matrix
from romtime_amd import ops
def penta(N, rng):
    offs = [-2, -1, 0, 1, 2]
    rows = np.concatenate([np.arange(max(0, -o), min(N, N - o)) for o in offs])
    cols = np.concatenate([np.arange(max(0, -o), min(N, N - o)) + o for o in offs])
    A = csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(N, N)); A.sort_indices(); return A
for N, r in [(300, 8), (300, 16), (300, 32), (300,48), (300, 80), (5000, 80), (5000, 128), (5000,96)]:
    rng = np.random.RandomState(1)
    A = penta(N, rng); V = rng.standard_normal((N, r))
    ip, ix = ops.to_device_index(A.indptr), ops.to_device_index(A.indices)
    AN = ops.project_csr(ip, ix, ops.to_device(A.data), ops.to_device(V)).cpu().numpy()
    ref = V.T @ (A @ V)
    err = np.abs(AN - ref) / np.abs(ref).max()
    T = (r + 15) // 16
    bad = [(i, j) for i in range(T) for j in range(T) if err[16*i:16*i+16, 16*j:16*j+16].max() > 1e-12]
    print(N, r, "max err", err.max(), "bad tiles", bad[:40], flush=True)
