"""Is the Gram kernel's clock power-limited?  Same launch on zeros / constant / random snapshots."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops
N, n = 1000000, 512
for name, X in (("random", torch.randn((N, n), dtype=torch.float64, device="cuda")),
                ("zeros", torch.zeros((N, n), dtype=torch.float64, device="cuda")),
                ("ones", torch.ones((N, n), dtype=torch.float64, device="cuda")),
                ("random again", torch.randn((N, n), dtype=torch.float64, device="cuda"))):
    for _ in range(5): ops.gram(X)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gram(X)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:14s} {ms:.3f} ms  {N * n * (n + 1) / ms / 1e9:.1f} TF", flush=True)
    del X
