"""The hyper-reduced step's expansion (64 x 280 times 280 x 6400) launched back to back through ops.gemm_nn, which routes
few-row / short-contraction / wide-output products to expansion_kernel (sweep.hip).  ROMTIME_SWEEP_FLAGS: 1 = generic GEMM,
0 = expansion kernel, 6 / 8 / 14 = its timing ablations (no global loads / one k-step / both; results wrong)."""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from romtime_amd import ops
G = torch.randn(64, 280, dtype=torch.float64, device="cuda")
Z = torch.randn(280, 6400, dtype=torch.float64, device="cuda")
out = torch.empty(64, 6400, dtype=torch.float64, device="cuda")
for _ in range(5): ops.gemm_nn(G, Z, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): ops.gemm_nn(G, Z, out=out)
e1.record(); torch.cuda.synchronize()
print("flags", os.environ.get("ROMTIME_SWEEP_FLAGS", "0"), "%%.2f us per launch" %% (e0.elapsed_time(e1) / 200 * 1e3), "err %%.1e" %% float((out - G @ Z).abs().max()), flush=True)
''' % os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for flags in ("1", "0", "6", "8", "14"):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROMTIME_SWEEP_FLAGS=flags), capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-400:], flush=True)
