"""A/B timing of the fused projection V^T (A_b V) under ROMTIME_PF_FLAGS variants (measurement tool, one GPU).

  python tools/probes/proj_ab.py [flags ...]      # default flags: 0

Shapes: C5's batched step (N = 1e5, 5 entries per row, 32 value vectors, r = 80), C4's projection (120 vectors, r = 80)
and a 120-mode basis (r = 120).  Prints the per-launch time of project_fused_kernel (the ctx's HIP event pair)
and the wall time of the whole call.
"""
import os, sys, subprocess

code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from scipy.sparse import csr_matrix
from romtime_amd import ops
from romtime_amd._lib import Context
def penta(N, rng):
    offs = [-2, -1, 0, 1, 2]
    rows = np.concatenate([np.arange(max(0, -o), min(N, N - o)) for o in offs])
    cols = np.concatenate([np.arange(max(0, -o), min(N, N - o)) + o for o in offs])
    A = csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(N, N)); A.sort_indices(); return A
rng = np.random.RandomState(0)
N = 100000
A = penta(N, rng)
ip, ix = ops.to_device_index(A.indptr), ops.to_device_index(A.indices)
ctx = Context.current()
out = []
for spec in os.environ.get("PF_SHAPES", "32x80,120x80,32x120,32x64,32x40").split(","):
    rowmajor = spec.endswith("r")
    B, r = map(int, spec.rstrip("r").split("x"))
    data = torch.randn((B, A.nnz), dtype=torch.float64, device="cuda")
    if rowmajor: data = data.T.contiguous().T   # (B, nnz) view of a C-ordered (nnz, B) array
    V, _ = torch.linalg.qr(torch.randn((N, r), dtype=torch.float64, device="cuda"))
    for _ in range(3): AN = ops.project_csr_batched(ip, ix, data.T, V)
    torch.cuda.synchronize()
    ctx.set_profile(True)
    ks = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        AN = ops.project_csr_batched(ip, ix, data.T, V)
    e1.record(); torch.cuda.synchronize()
    for _ in range(5):
        AN = ops.project_csr_batched(ip, ix, data.T, V); torch.cuda.synchronize(); ks.append(ctx.last_gemm_ms())
    ctx.set_profile(False)
    flops = B * (2.0 * A.nnz * r + 2.0 * N * r * r)
    Ad = torch.sparse_csr_tensor(ip, ix, data[1], size=(N, N))
    ref = V.T @ (Ad @ V)
    err = float((AN[1] - ref).abs().max() / ref.abs().max())
    k = float(np.median(ks))
    out.append(("rm " if rowmajor else "") + "B%%d r%%d: kernel %%.4f ms (%%.3f of 78.6 TF) call %%.4f ms err %%.1e" %% (B, r, k, flops / k / 1e9 / 78.6, e0.elapsed_time(e1) / 10, err))
print("flags", os.environ.get("ROMTIME_PF_FLAGS", "0"), " | ".join(out))
''' % os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

flags = sys.argv[1:] or ["0"]
for rep in range(2):
    for f in flags:
        o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROMTIME_PF_FLAGS=f), capture_output=True, text=True)
        print(o.stdout.strip() or o.stderr[-600:], flush=True)
