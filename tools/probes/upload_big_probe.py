"""Host -> device upload of one large pageable NumPy snapshot matrix (the drop-in orth(ndarray) path): torch's pageable
copy against a chunked copy through two pinned buffers filled by a few host threads.
python3 tools/probes/upload_big_probe.py [rows] [cols]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
X = np.random.default_rng(0).standard_normal((N, n))
gb = X.nbytes / 1e9
torch.cuda.init()
torch.zeros(1, device="cuda")

for rep in range(2):
    t0 = time.perf_counter()
    D = ops.to_device(X)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"ops.to_device (pageable): {dt * 1e3:7.1f} ms  {gb / dt:5.1f} GB/s", flush=True)
    ref = D
    del D


def chunked(X, chunk_rows, threads):
    dev = torch.empty(X.shape, dtype=torch.float64, device="cuda")
    bufs = [torch.empty((chunk_rows, X.shape[1]), dtype=torch.float64).pin_memory() for _ in range(2)]
    free = [None, None]
    pool = ThreadPoolExecutor(threads)
    t0 = time.perf_counter()
    for ci, r0 in enumerate(range(0, X.shape[0], chunk_rows)):
        r1 = min(X.shape[0], r0 + chunk_rows)
        b = ci & 1
        if free[b] is not None:
            free[b].synchronize()
        host = bufs[b].numpy()[: r1 - r0]
        step = (r1 - r0 + threads - 1) // threads
        list(pool.map(lambda k: np.copyto(host[k * step:(k + 1) * step], X[r0 + k * step:min(r1, r0 + (k + 1) * step)]), range(threads)))
        dev[r0:r1].copy_(bufs[b][: r1 - r0], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        free[b] = ev
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pool.shutdown()
    return dev, dt


for chunk_rows, threads in [(65536, 1), (65536, 4), (65536, 8), (32768, 8), (131072, 8), (65536, 12)]:
    dev, dt = chunked(X, chunk_rows, threads)
    dev2, dt2 = chunked(X, chunk_rows, threads)
    print(f"pinned double buffer, {chunk_rows} rows per chunk, {threads} host threads: {dt * 1e3:7.1f} / {dt2 * 1e3:7.1f} ms  {gb / dt2:5.1f} GB/s"
          f"  equal {bool(torch.equal(dev2, ref))}", flush=True)
    del dev, dev2
