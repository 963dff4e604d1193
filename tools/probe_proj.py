import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romtime_amd import ops
from tools.bench_configs import penta, wall_time
N, r, B = 100_000, 80, 120
rng = np.random.RandomState(4)
A = penta(N, rng)
ip, ix = ops.to_device_index(A.indptr), ops.to_device_index(A.indices)
V, _ = torch.linalg.qr(torch.randn((N, r), dtype=torch.float64, device="cuda"))
Phi = torch.randn((B, A.nnz), dtype=torch.float64, device="cuda").T   # col-major (each mode contiguous)
ms, AN = wall_time(lambda: ops.project_csr_batched(ip, ix, Phi, V), reps=5)
print("project fused ms", ms, "TF", B * (2.0 * A.nnz * r + 2.0 * N * r * r) / ms / 1e9)
