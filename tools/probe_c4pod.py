"""Timeline of one deflated POD (C4's 5e5 x 200 snapshot set): run under rocprofv3 --kernel-trace and print the
kernels of the last pod_device call in launch order with start offsets and durations."""
import numpy as np
import torch

from romtime_amd import pod

rng = np.random.RandomState(4)
nnz, n_ops = 499_994, 200
x = np.linspace(0, 1, nnz)
B = np.stack([np.sin((q + 1) * np.pi * x) * (1 + 0.1 * q) for q in range(8)], axis=1)
S = torch.from_numpy(B @ rng.standard_normal((8, n_ops)) + 1e-6 * rng.standard_normal((nnz, n_ops))).cuda()
S[0, :] = 0.0
for _ in range(3):
    out = pod.pod_device(S, num=120, normalize=False)
torch.cuda.synchronize()
mark = torch.zeros(12345, device="cuda")   # a recognisable fill kernel marks the start of the traced call
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
out = pod.pod_device(S, num=120, normalize=False)
torch.cuda.synchronize()
print("wall ms", 1e3 * (time.perf_counter() - t0), "levels/passes", out["passes"], "r", out["r"])
