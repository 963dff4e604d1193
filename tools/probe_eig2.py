import time, torch, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romtime_amd import ops
n = 512
rng = np.random.RandomState(0)
s = 10.0 ** (-8 * np.arange(n) / (n - 1)); V, _ = np.linalg.qr(rng.standard_normal((n, n))); G = (V * s**2) @ V.T; G = (G + G.T) / 2
Gd = ops.to_device(G)
def T(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); out = fn(); torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t), out
for i in range(14):
    t1, (lam, st) = T(lambda: ops.sym_eig_values(Gd))
    t2, W = T(lambda: ops.sym_eig_vectors(lam, 40))
    t3, GZ = T(lambda: ops.gemm_nn(Gd, W))
    t4, h = T(lambda: lam.cpu())
    print(f"it {i}: values {t1:.2f} vectors {t2:.2f} gemm {t3:.3f} d2h {t4:.3f}  status {int(st.item())}")
ref = np.linalg.eigvalsh(G)[::-1]
print("max abs err", np.abs(lam.cpu().numpy() - ref).max())
