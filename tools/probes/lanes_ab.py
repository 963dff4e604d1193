"""PodLanes throughput on 32 sets of 1e5 x 256 against the number of lanes (and GPU_MAX_HW_QUEUES from the environment)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import pod, pipeline
N, n, r, sets = 100_000, 256, 40, 32
g = torch.Generator(device="cuda").manual_seed(2)
s = torch.from_numpy(10.0 ** (-6.0 * np.arange(n) / (n - 1))).cuda()
Xs = []
for _ in range(sets):
    V0, _ = torch.linalg.qr(torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g))
    Xs.append((torch.randn((N, n), dtype=torch.float64, device="cuda", generator=g) / np.sqrt(N)) @ (s[:, None] * V0.T))
def wall(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps
print("queues", os.environ.get("GPU_MAX_HW_QUEUES", "default"), "sequential ms per POD %.3f" % (wall(lambda: [pod.pod_device(X, num=r, normalize=True) for X in Xs]) / sets), flush=True)
for L in (1, 2, 4, 8):
    lanes = pipeline.PodLanes(lanes=L)
    print("lanes", L, "ms per POD %.3f" % (wall(lambda: lanes.map(Xs, num=r, normalize=True)) / sets), "recomputed", lanes.recomputed, flush=True)
