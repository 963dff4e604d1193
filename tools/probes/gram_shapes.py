"""Time of the snapshot Gram kernel over a few shapes (ROMTIME_GRAM_FLAGS / ROMTIME_GRAM_PACE from the environment) and
its error against torch on a slice.   python3 tools/probes/gram_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops  # noqa: E402
from romtime_amd._lib import Context  # noqa: E402

tag = os.environ.get("ROMTIME_GRAM_FLAGS", "-") + ":" + os.environ.get("ROMTIME_GRAM_PACE", "-")
SHORT = [(100_000, 256, "C"), (100_000, 200, "C"), (50_000, 512, "C"), (30_000, 384, "C"), (200_000, 256, "C"), (100_000, 384, "C"),
         (60_000, 1000, "C"), (100_000, 256, "F"), (250_000, 200, "C")]
for (N, n, order) in SHORT if "short" in sys.argv else [(1_000_000, 512, "C"), (1_000_000, 384, "C"), (1_000_000, 256, "C"), (600_000, 640, "C"),
                      (500_000, 768, "C"), (400_000, 1024, "C"), (1_000_000, 500, "C"), (300_000, 512, "C"),
                      (1_000_000, 512, "F")]:
    X = torch.randn((N, n) if order == "C" else (n, N), dtype=torch.float64, device="cuda")
    if order == "F":
        X = X.T
    for _ in range(3):
        G = ops.gram(X)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        G = ops.gram(X)
    e1.record()
    torch.cuda.synchronize()
    ref = X[:, :8].T @ X
    info = Context.current().launch_info()
    print(f"{tag} N={N} n={n} {order}: {e0.elapsed_time(e1) / 20:.3f} ms  grid {info['grid']} splits {info['splits']}  "
          f"err {float((G[:8] - ref).abs().max() / ref.abs().max()):.1e}  sym {bool(torch.equal(G, G.T))}", flush=True)
    del X, G, ref
