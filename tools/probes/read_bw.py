"""How fast can this chip READ a 1e6 x 512 float64 matrix (4.1 GB) once?  torch reductions as streaming-read kernels, and
the library's own copy benchmark; the back-projection's 1.04 ms = 4.25 TB/s is to be read against these."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
X = torch.randn((1_000_000, 512), dtype=torch.float64, device="cuda")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = X.numel() * 8 / 1e9
for name, fn in (("X.sum()", lambda: X.sum()), ("X.sum(dim=0)", lambda: X.sum(dim=0)), ("X.abs().max()", lambda: X.abs().max()),
                 ("torch.linalg.vector_norm", lambda: torch.linalg.vector_norm(X))):
    ms = timeit(fn)
    print(f"{name:28s} {ms:.3f} ms  {gb / ms:.2f} TB/s", flush=True)
Y = torch.empty_like(X)
ms = timeit(lambda: Y.copy_(X))
print(f"{'copy (read + write)':28s} {ms:.3f} ms  {2 * gb / ms:.2f} TB/s moved", flush=True)
