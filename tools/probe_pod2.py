import time, torch, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from romtime_amd import ops
X = bench.build_local_matrix(0, 1, 1_000_000, 512, torch.device("cuda", 0))
def T(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); out = fn(); torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t), out
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
for i in range(10):
    t0, G = T(lambda: ops.gram(X))
    t1, (cn, fl) = T(lambda: ops.gram_scale(G, True))
    t2, (lam, st) = T(lambda: ops.sym_eig_values(G))
    t3, Z = T(lambda: ops.sym_eig_vectors(lam, 40))
    t4, Tm = T(lambda: ops.gemm_nn(Z, torch.eye(40, dtype=torch.float64, device="cuda")))
    t5, Q = T(lambda: ops.gemm_nn(X, Tm))
    print(f"it {i}: gram {t0:.2f} scale {t1:.2f} eigvals {t2:.2f} eigvecs {t3:.2f} small {t4:.2f} backproj {t5:.2f}")
