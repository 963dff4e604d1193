"""n <= 256 instantiations of the eigensolver kernels (round 3) against the 512 ones (ROMTIME_EIG_FLAGS=2): values + 40
vectors at n = 64 ... 256, ms per call.  python tools/probes/eig_small_ab.py on the GPU box."""
import os, subprocess, sys
code = r'''
import os, sys, time, numpy as np, torch
sys.path.insert(0, %r)
from romtime_amd import ops
for n in (64, 128, 200, 256):
    rng = np.random.RandomState(0)
    s = 10.0 ** (-6 * np.arange(n) / (n - 1))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    G = (V * s ** 2) @ V.T; G = (G + G.T) / 2
    Gd = torch.from_numpy(G).cuda()
    def ours():
        lam, _ = ops.sym_eig_values(Gd)
        return lam, ops.sym_eig_vectors(lam, min(40, n))
    for _ in range(3): ours()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): lam, W = ours()
    torch.cuda.synchronize(); ms = (time.time() - t) / 20 * 1e3
    lam, W = lam.cpu().numpy(), W.cpu().numpy()
    res = np.abs(G @ W - W * lam[:W.shape[1]]).max() / lam[0]
    print("flags", os.environ.get("ROMTIME_EIG_FLAGS", "0"), "n", n, "ms %%.3f" %% ms, "residual %%.1e" %% res, flush=True)
''' % os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for flags in ("2", "0", "2", "0"):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROMTIME_EIG_FLAGS=flags), capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-500:], flush=True)
