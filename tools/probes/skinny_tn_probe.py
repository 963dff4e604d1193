"""C = A^T B with few columns of A (the deflation sweep's first half): streaming kernel (rank_update.hip) against the
generic GEMM (ROMTIME_DEFLATE_FLAGS=1), correctness against torch and microseconds per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops  # noqa: E402

torch.manual_seed(0)
print("ROMTIME_DEFLATE_FLAGS =", os.environ.get("ROMTIME_DEFLATE_FLAGS", "0"))
for (N, n, m, pad) in [(1_000_000, 200, 4, 0), (1_000_000, 200, 8, 0), (1_000_000, 200, 16, 0), (1_000_000, 512, 8, 0),
                       (1_000_000, 512, 16, 0), (100_000, 256, 10, 0), (300_001, 201, 7, 3), (65_537, 77, 1, 0),
                       (1_000_000, 200, 24, 0)]:
    B = torch.randn(N, n + pad, dtype=torch.float64, device="cuda")[:, :n]
    Afull = torch.randn(N, 40, dtype=torch.float64, device="cuda")
    A = Afull[:, :m]
    C = ops.gemm_tn(A, B)
    ref = A.T @ B
    err = float((C - ref).abs().max() / ref.abs().max())
    C2 = ops.gemm_tn(A, B)
    same = bool(torch.equal(C, C2))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.gemm_tn(A, B)
    e0.record()
    reps = 20
    for _ in range(reps):
        ops.gemm_tn(A, B)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"N={N} n={n} m={m} pad={pad}: rel err {err:.2e} repeatable {same} {us:8.1f} us  {8.0 * N * n / us / 1e6:6.2f} TB/s")
    del B, Afull, A
