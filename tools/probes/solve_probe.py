"""newton_solve_kernel on its own: 32 systems of 80, K drifting by 1e-4 per call (the online sweep's situation), the
inverse carried from call to call.  Under rocprofv3 --kernel-trace the kernel's average is its device time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops
from romtime_amd._lib import Context
torch.manual_seed(0)
B, r = 32, 80
A = torch.randn(B, r, r, dtype=torch.float64, device="cuda")
K0 = A @ A.transpose(1, 2) / r + 3.0 * torch.eye(r, dtype=torch.float64, device="cuda")
D = torch.randn(B, r, r, dtype=torch.float64, device="cuda")
b = torch.randn(B, r, dtype=torch.float64, device="cuda")
Xinv = None
worst = 0.0
for step in range(300):
    K = K0 + 1e-4 * step * D
    x, info, Xinv = ops.tracked_solve(K, b, Xinv)
    if step % 50 == 0:
        worst = max(worst, float(((K @ x.unsqueeze(-1)).squeeze(-1) - b).norm() / b.norm()))
torch.cuda.synchronize()
print("residual", worst, Context.current().sweep_stats() if False else "")
