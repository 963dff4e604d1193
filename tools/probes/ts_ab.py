import os, sys, subprocess, json
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(%r)))))
from romtime_amd import ops
X = torch.randn((1000000, 512), dtype=torch.float64, device="cuda")
T = torch.randn((512, 40), dtype=torch.float64, device="cuda")
for _ in range(5): ops.gemm_nn(X, T)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): Y = ops.gemm_nn(X, T)
e1.record(); torch.cuda.synchronize()
ref = X[:4096] @ T
print("flags", os.environ.get("ROMTIME_TS_FLAGS", "0"), "ms", e0.elapsed_time(e1) / 30, "err", float((Y[:4096] - ref).abs().max()))
''' % __file__
for rep in range(2):
    for flags in ("0", "4", "8", "16", "24"):   # 0 = staged kernel, 4 = direct-operand kernel, 8 = non-temporal X loads, 16 = non-temporal Y stores
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROMTIME_TS_FLAGS=flags), capture_output=True, text=True)
        print(out.stdout.strip() or out.stderr[-300:], flush=True)
