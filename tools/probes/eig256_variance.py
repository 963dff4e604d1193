"""Why does values + 40 vectors at n = 256 take 0.89 ms in most processes and 1.3-1.9 ms in some?  Ten processes, each:
ms per call over 20 calls, and which hand-off form the tridiagonalisations took (rt_ctx_get_counter)."""
import os, subprocess, sys
code = r'''
import os, sys, time, numpy as np, torch
sys.path.insert(0, %r)
from romtime_amd import ops
from romtime_amd._lib import Context
n = 256
rng = np.random.RandomState(0)
s = 10.0 ** (-6 * np.arange(n) / (n - 1))
V, _ = np.linalg.qr(rng.standard_normal((n, n)))
G = (V * s ** 2) @ V.T; G = (G + G.T) / 2
Gd = torch.from_numpy(G).cuda()
def ours():
    lam, _ = ops.sym_eig_values(Gd)
    return lam, ops.sym_eig_vectors(lam, 40)
for _ in range(3): ours()
torch.cuda.synchronize()
per = []
for _ in range(20):
    t = time.time(); ours(); torch.cuda.synchronize(); per.append((time.time() - t) * 1e3)
ctx = Context.current()
print("ms min %%.3f median %%.3f max %%.3f" %% (min(per), sorted(per)[10], max(per)),
      "one_xcd", ctx.counter("eig_one_xcd"), "general", ctx.counter("eig_general_form"), "timeouts", ctx.counter("eig_timeouts"), flush=True)
''' % os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rep in range(10):
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-500:], flush=True)
