"""One shape of the fused projection, a few launches: the target of rocprofv3 --pmc runs.
   python3 tools/probes/proj_one.py [B] [r] [N]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.sparse import csr_matrix
from romtime_amd import ops
B, r, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), (int(sys.argv[2]) if len(sys.argv) > 2 else 80), (int(sys.argv[3]) if len(sys.argv) > 3 else 100000)
offs = [-2, -1, 0, 1, 2]
rows = np.concatenate([np.arange(max(0, -o), min(N, N - o)) for o in offs])
cols = np.concatenate([np.arange(max(0, -o), min(N, N - o)) + o for o in offs])
A = csr_matrix((np.random.RandomState(0).standard_normal(rows.size), (rows, cols)), shape=(N, N)); A.sort_indices()
ip, ix = ops.to_device_index(A.indptr), ops.to_device_index(A.indices)
data = torch.randn((B, A.nnz), dtype=torch.float64, device="cuda")
V, _ = torch.linalg.qr(torch.randn((N, r), dtype=torch.float64, device="cuda"))
for _ in range(6): AN = ops.project_csr_batched(ip, ix, data.T, V)
torch.cuda.synchronize()
print("done", float(AN.abs().sum()))
