"""The Gram of a 1e6 x 512 snapshot matrix, a few launches: time per Gram (ROMTIME_GRAM_FLAGS from the environment), and
the target of `rocprofv3 --pmc FETCH_SIZE` runs that count its HBM-side reads.
   python3 tools/probes/gram_traffic_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops
N, n = 1000000, 512
X = torch.randn((N, n), dtype=torch.float64, device="cuda")
for _ in range(3): G = ops.gram(X)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): G = ops.gram(X)
e1.record(); torch.cuda.synchronize()
ref = X[:, :8].T @ X
print("flags", os.environ.get("ROMTIME_GRAM_FLAGS", "1"), "ms per Gram %.3f" % (e0.elapsed_time(e1) / 10), "err %.1e" % float((G[:8] - ref).abs().max() / ref.abs().max()))
