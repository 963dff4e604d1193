"""Quick timing probe for the Gram / GEMM kernels (not part of the test suite)."""
import sys, time
import torch
from romtime_amd import ops
from romtime_amd._lib import Context

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

print("mfma f64 peak TF:", ops.bench_mfma_f64(20000))
print("copy GB/s:", ops.bench_copy(1 << 30, 10))
for (N, n) in [(100000, 256), (1000000, 512), (1000000, 64)]:
    for order in ("C", "F"):
        X = torch.randn((N, n) if order == "C" else (n, N), dtype=torch.float64, device="cuda")
        if order == "F": X = X.T
        ms = timeit(lambda: ops.gram(X))
        fl = N * n * (n + 1)
        print(f"gram {N}x{n} {order}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF(alg)  {N*n*8/ms/1e6:.0f} GB/s  info={Context.current().launch_info()}")
        if n >= 256:
            T = torch.randn((n, 40), dtype=torch.float64, device="cuda")
            ms = timeit(lambda: ops.gemm_nn(X, T))
            print(f"   gemm_nn k=40: {ms:.3f} ms {2*N*n*40/ms/1e9:.1f} TF  {N*(n+40)*8/ms/1e6:.0f} GB/s")
        del X
