import time, torch, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from romtime_amd import pod
from romtime_amd._lib import Context
X = bench.build_local_matrix(0, 1, 1_000_000, 512, torch.device("cuda", 0))
ctx = Context.current()
for prof in (False, True):
    ctx.set_profile(prof)
    for i in range(3): pod.pod_device(X, num=40, normalize=True)
    ts = []
    for i in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = pod.pod_device(X, num=40, normalize=True)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print("profile", prof, " ".join(f"{t:.1f}" for t in ts))
