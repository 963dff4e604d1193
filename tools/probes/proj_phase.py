"""Phase relation of the two workgroups a CU holds (ablation build only: ROMTIME_EXTRA_HIPFLAGS=-DROMTIME_PF_ABLATE).
Reads the s_memtime stamps project_fused_kernel leaves around its MFMA phases and prints, per CU, how the
MFMA phases of its workgroups lie relative to each other."""
import os, sys, ctypes, numpy as np, torch
os.environ["ROMTIME_PF_FLAGS"] = str(16384 + int(sys.argv[1]) if len(sys.argv) > 1 else 16384)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.sparse import csr_matrix
from romtime_amd import ops
from romtime_amd._lib import Context
B, r, N = 32, 80, 100000
offs = [-2, -1, 0, 1, 2]
rows = np.concatenate([np.arange(max(0, -o), min(N, N - o)) for o in offs])
cols = np.concatenate([np.arange(max(0, -o), min(N, N - o)) + o for o in offs])
A = csr_matrix((np.random.RandomState(0).standard_normal(rows.size), (rows, cols)), shape=(N, N)); A.sort_indices()
ip, ix = ops.to_device_index(A.indptr), ops.to_device_index(A.indices)
data = torch.randn((B, A.nnz), dtype=torch.float64, device="cuda")
V, _ = torch.linalg.qr(torch.randn((N, r), dtype=torch.float64, device="cuda"))
for _ in range(3): AN = ops.project_csr_batched(ip, ix, data.T, V)
torch.cuda.synchronize()
lib = Context.current().lib
n = 1024 * 20
buf = (ctypes.c_ulonglong * n)()
assert lib.rt_pf_debug_dump(buf, n) == 0
d = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 20).astype(np.int64)
hw = d[:, 0]
cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | ((hw >> 32) << 8)
st = d[:, 2:18].reshape(1024, 8, 2)            # [wg][stage][head, tail]
print("MFMA phase length (cycles of s_memtime):", np.median(st[:, :, 1] - st[:, :, 0]), " stage period:", np.median(np.diff(st[:, :, 0], axis=1)))
from collections import defaultdict
by = defaultdict(list)
for w in range(1024): by[int(cu[w])].append(w)
print("CUs seen:", len(by), " workgroups per CU:", sorted(set(len(v) for v in by.values())))
ov = []
for c, ws in list(by.items()):
    ws = sorted(ws, key=lambda w: st[w, 0, 0])
    # first-round pair: the two earliest workgroups
    a, b = ws[0], ws[1]
    per = np.median(np.diff(st[a, :, 0]))
    off = (st[b, :, 0] - st[a, :, 0]) % per
    ov.append(np.median(off) / per)
ov = np.array(ov)
print("offset of the second workgroup's MFMA phase within the first's stage period (fraction): quartiles", np.percentile(ov, [10, 25, 50, 75, 90]).round(2))
