import time, torch, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from romtime_amd import ops
X = bench.build_local_matrix(0, 1, 1_000_000, 512, torch.device("cuda", 0))
E = lambda: torch.cuda.Event(enable_timing=True)
Tm = torch.randn(512, 40, dtype=torch.float64, device="cuda")
for i in range(9):
    ev = [E() for _ in range(8)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev[0].record(); G = ops.gram(X)
    ev[1].record(); cn, fl = ops.gram_scale(G, True)
    ev[2].record(); lam, st = ops.sym_eig_values(G)
    ev[3].record(); head = torch.cat([lam, st.to(torch.float64), fl.to(torch.float64)]).cpu()
    th = 1e3 * (time.perf_counter() - t0)
    ev[4].record(); Z = ops.sym_eig_vectors(lam, 40)
    ev[5].record(); Q = ops.gemm_nn(X, Tm)
    ev[6].record()
    torch.cuda.synchronize(); tt = 1e3 * (time.perf_counter() - t0)
    d = [ev[j].elapsed_time(ev[j + 1]) for j in range(6)]
    print(f"it {i}: wall {tt:.1f} host_to_d2h {th:.1f} | gram {d[0]:.2f} scale {d[1]:.2f} eigvals {d[2]:.2f} cat+d2h {d[3]:.2f} vecs {d[4]:.2f} backproj {d[5]:.2f}")
