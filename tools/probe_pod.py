import time, torch, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from romtime_amd import pod
from romtime_amd._lib import Context
X = bench.build_local_matrix(0, 1, 1_000_000, 512, torch.device("cuda", 0))
ctx = Context.current()
for i in range(3): pod.pod_device(X, num=40, normalize=True)
ctx.set_profile(True)
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pod.pod_device(X, num=40, normalize=True)
    torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
    print(f"step {i}: wall {dt:.2f} ms", {k: round(v, 2) for k, v in pod.stage_timings().items()})
print("---- allocator probe")
for i in range(7):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pod.pod_device(X, num=40, normalize=True)
    torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
    st1 = torch.cuda.memory_stats()
    print(f"step {i}: wall {dt:.2f} ms dev_alloc {st1['num_device_alloc']-st0['num_device_alloc']} dev_free {st1['num_device_free']-st0['num_device_free']} reserved {st1['reserved_bytes.all.current']/2**20:.0f} MiB")
