"""Device symmetric eigensolver (values + 40 vectors) against rocSOLVER (torch.linalg.eigh) and host LAPACK."""
import time

import numpy as np
import torch

from romtime_amd import ops

for n in (256, 512, 768, 1024):
    rng = np.random.RandomState(0)
    s = 10.0 ** (-8 * np.arange(n) / (n - 1))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    G = (V * s ** 2) @ V.T
    G = (G + G.T) / 2
    Gd = torch.from_numpy(G).cuda()

    def ours():
        lam, _ = ops.sym_eig_values(Gd)
        return ops.sym_eig_vectors(lam, 40)

    # rocSOLVER only when asked for (python tools/probe_eig.py all): timing it in the same process perturbs the
    # timings that follow it
    import sys
    cands = [("romtime_amd values + 40 vectors", ours)]
    if "all" in sys.argv[1:]:
        cands.append(("torch.linalg.eigh (rocSOLVER)", lambda: torch.linalg.eigh(Gd)))
    for name, fn in cands:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        print(f"n={n:5d} {name}: {(time.time() - t) / 5 * 1e3:.2f} ms", flush=True)
    t = time.time()
    np.linalg.eigh(G)
    print(f"n={n:5d} host numpy eigh: {(time.time() - t) * 1e3:.1f} ms", flush=True)
