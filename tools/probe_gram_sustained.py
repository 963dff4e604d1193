"""Does the Gram kernel slow down when it runs for longer than a short burst (power / clock management)?"""
import torch
from romtime_amd import ops
X = torch.randn((1_000_000, 512), dtype=torch.float64, device="cuda")
ops.gram(X); torch.cuda.synchronize()
for reps in (3, 10, 30, 100):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ops.gram(X)
    e1.record(); torch.cuda.synchronize()
    print(f"{reps:4d} back-to-back Grams: {e0.elapsed_time(e1)/reps:.3f} ms each", flush=True)
