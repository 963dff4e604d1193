import os, sys, torch
from romtime_amd import ops
from romtime_amd._lib import Context
def timeit(fn, reps=8):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N, n = 1000000, 512
X = torch.randn((N, n), dtype=torch.float64, device="cuda")
for rep in range(2):
    ms = timeit(lambda: ops.gram(X))
    print(f"flags={os.environ.get('ROMTIME_GRAM_FLAGS','0')} gram {N}x{n}: {ms:.3f} ms  {N*n*(n+1)/ms/1e9:.1f} TF(alg)")
