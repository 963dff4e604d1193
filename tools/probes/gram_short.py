import os, sys, torch
sys.path.insert(0, os.getcwd())
from romtime_amd import ops
for N, n in ((100000, 256), (100000, 200), (50000, 512), (200000, 128), (30000, 384)):
    X = torch.randn((N, n), dtype=torch.float64, device="cuda")
    for _ in range(3): G = ops.gram(X)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): G = ops.gram(X)
    e1.record(); torch.cuda.synchronize()
    ref = X.T @ X
    print("short", os.environ.get("ROMTIME_GRAM_SHORT", "0"), (N, n), "us %.1f" % (e0.elapsed_time(e1) / 20 * 1e3), "err %.1e" % float((G - ref).abs().max() / ref.abs().max()), flush=True)
