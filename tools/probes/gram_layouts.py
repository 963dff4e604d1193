import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops
for name, X in (("row-major 1e6x512", torch.randn((1000000, 512), dtype=torch.float64, device="cuda")),
                ("col-major 1e6x512", torch.randn((512, 1000000), dtype=torch.float64, device="cuda").T)):
    for _ in range(3): G = ops.gram(X)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): G = ops.gram(X)
    e1.record(); torch.cuda.synchronize()
    ref = X[:20000].T @ X[:20000]
    G2 = ops.gram(X[:20000] if X.stride(1) == 1 else X[:20000])
    print(name, "ms", e0.elapsed_time(e1) / 10, "rel err (20000 rows)", float((G2 - ref).abs().max() / ref.abs().max()))
