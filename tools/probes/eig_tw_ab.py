"""Team size of the tridiagonalisation for small n (ROMTIME_EIG_TW): values + 40 vectors, ms per call."""
import os, subprocess, sys
code = r'''
import os, sys, time, numpy as np, torch
sys.path.insert(0, %r)
from romtime_amd import ops
for n in (16, 33, 64, 100, 128, 256):
    rng = np.random.RandomState(0)
    s = 10.0 ** (-6 * np.arange(n) / (n - 1))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    G = (V * s ** 2) @ V.T; G = (G + G.T) / 2
    Gd = torch.from_numpy(G).cuda()
    def ours():
        lam, _ = ops.sym_eig_values(Gd)
        return lam, ops.sym_eig_vectors(lam, min(40, n))
    for _ in range(5): ours()
    torch.cuda.synchronize(); per = []
    for _ in range(20):
        t = time.time(); lam, W = ours(); torch.cuda.synchronize(); per.append((time.time() - t) * 1e3)
    lam, W = lam.cpu().numpy(), W.cpu().numpy()
    res = np.abs(G @ W - W * lam[:W.shape[1]]).max() / lam[0]
    print("tw", os.environ.get("ROMTIME_EIG_TW", "32"), "n", n, "median ms %%.3f" %% sorted(per)[10], "residual %%.1e" %% res, flush=True)
''' % os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for flags in ("4", "0", "4", "0"):   # 4 = cooperative form also for n <= 128; 0 = the single-workgroup form there
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROMTIME_EIG_FLAGS=flags), capture_output=True, text=True)
    print("EIG_FLAGS", flags)
    print(out.stdout.strip() or out.stderr[-500:], flush=True)
