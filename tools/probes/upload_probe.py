import time, numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from romtime_amd import ops
N, n = 100_000, 256
vecs = [np.random.rand(N) for _ in range(n)]
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); S = np.array(vecs).T; t1 = time.perf_counter()
    X = ops.to_device(S); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"stack {1e3*(t1-t0):.1f} ms  upload(pageable) {1e3*(t2-t1):.1f} ms", flush=True)
pin = torch.empty((n, N), dtype=torch.float64).pin_memory()
dev = torch.empty((n, N), dtype=torch.float64, device="cuda")
for rep in range(3):
    t0 = time.perf_counter(); np.stack(vecs, out=pin.numpy()); t1 = time.perf_counter()
    dev.copy_(pin, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"stack into pinned {1e3*(t1-t0):.1f} ms  dma {1e3*(t2-t1):.1f} ms", flush=True)
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(8)
src = np.array(vecs)
def cp(i):
    a, b = i * n // 8, (i + 1) * n // 8
    np.copyto(pin.numpy()[a:b], src[a:b])
for rep in range(3):
    t0 = time.perf_counter(); list(pool.map(cp, range(8))); t1 = time.perf_counter()
    print(f"8-thread copy into pinned {1e3*(t1-t0):.1f} ms", flush=True)
